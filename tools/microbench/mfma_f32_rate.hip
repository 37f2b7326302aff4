// Bare v_mfma_f32_32x32x2_f32 rate: W waves per SIMD, A accumulators per wave, operands in registers (random), 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int A>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
  f32x16 acc[A];
  float a = in[threadIdx.x], b = in[threadIdx.x + 512];
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < A; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < A; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int A>
void run(int threads, int iters, float* out, float* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<A>, dim3(256), dim3(threads), 0, 0, out, in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<A>, dim3(256), dim3(threads), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 256.0 * (threads / 64) * iters * 8.0 * A * (32.0 * 32 * 2 * 2);
  printf("waves/SIMD %d accs %d: %.2f ms  %.1f TFLOP/s\n", threads / 256, A, ms, flops / ms / 1e9);
}
int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 1024 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<1>(256, 20000, out, in); run<2>(256, 20000, out, in); run<4>(256, 20000, out, in);
  run<1>(512, 20000, out, in); run<2>(512, 20000, out, in); run<4>(512, 20000, out, in);
  return 0;
}
