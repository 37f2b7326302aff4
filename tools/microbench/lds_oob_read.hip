// Does a ds_read_b128 beyond the workgroup's LDS allocation return zeros on gfx950?  (The border taps of the 3x3 kernels
// could then be masked by ONE address add instead of a select between the pixel's row and a row of zeros.)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_oob_read.hip -o /tmp/lds_oob_read && /tmp/lds_oob_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, unsigned far) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(sm)[i] = 1.0f + i;
  __syncthreads();
  const unsigned lane = threadIdx.x;
  unsigned addr = lane * 16u;
  if (lane & 1) addr += far;                    // odd lanes: beyond the allocation
  typedef __attribute__((address_space(3))) f32x4* lp;
  const f32x4 v = *reinterpret_cast<lp>((unsigned long long)addr + 2048);   // with an immediate offset on top
  out[lane * 4 + 0] = v[0]; out[lane * 4 + 1] = v[1]; out[lane * 4 + 2] = v[2]; out[lane * 4 + 3] = v[3];
}
int main() {
  float* out; (void)hipMalloc(&out, 64 * 4 * 4);
  const unsigned fars[] = {1u << 20, 1u << 18, 0x80000000u, 65536u};
  for (unsigned far : fars) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, out, far);
    float h[256];
    if (hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("far %#x: launch failed\n", far); return 1; }
    int bad_even = 0, nonzero_odd = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        const float v = h[l * 4 + e];
        if (l & 1) nonzero_odd += v != 0.f;
        else bad_even += v != 1.0f + (l * 16 + 2048) / 4 + e;
      }
    printf("far %#x: in-range lanes wrong %d, out-of-range lanes non-zero %d (of 128)  sample odd lane: %g %g\n", far, bad_even, nonzero_odd, h[4], h[5]);
  }
  return 0;
}
