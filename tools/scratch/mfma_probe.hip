// scratch probe: what limits v_mfma_f32_32x32x2_f32 throughput on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: pure MFMA (registers only), 4 accumulators
// MODE 1: + ds_read_b128 of operands each 8-k group (LDS resident, like the conv inner loop)
// MODE 2: like 1 with 2 accumulators only (1x2 tile)
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 2 * 128 * 36; i += blockDim.x) lds[i] = (float)(i % 7) * 0.001f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 fa[2], fb[2];
  fa[0] = fa[1] = fb[0] = fb[1] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
  const float* as = lds + r * 36 + 4 * h;
  const float* bs = lds + 128 * 36 + r * 36 + 4 * h;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE >= 1) {
        fa[0] = *reinterpret_cast<const float4*>(as + 8 * j);
        fa[1] = *reinterpret_cast<const float4*>(as + 32 * 36 + 8 * j);
        fb[0] = *reinterpret_cast<const float4*>(bs + 8 * j);
        fb[1] = *reinterpret_cast<const float4*>(bs + 32 * 36 + 8 * j);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[jn].x, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[jn].y, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[jn].z, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[jn].w, acc[i][jn], 0, 0, 0);
        }
    }
    if (MODE == 0) { asm volatile("" : "+v"(fa[0].x), "+v"(fb[0].x)); }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES>
void run(const char* name, int blocks_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  double flops = (double)grid * WAVES * iters * 64 * 4096.0;
  printf("%-34s grid %5d x %d waves: %8.1f us  %6.1f TF\n", name, grid, WAVES, best * 1e3, flops / best / 1e9);
  hipFree(out);
}

int main() {
  run<0, 4>("pure MFMA, 1 wave/SIMD", 1);
  run<0, 4>("pure MFMA, 2 waves/SIMD", 2);
  run<0, 4>("pure MFMA, 4 waves/SIMD", 4);
  run<1, 4>("MFMA + ds_read_b128, 1 wave/SIMD", 1);
  run<1, 4>("MFMA + ds_read_b128, 2 waves/SIMD", 2);
  run<1, 4>("MFMA + ds_read_b128, 4 waves/SIMD", 4);
  run<0, 1>("pure MFMA, 1-wave WGs x4/CU", 4);
  run<0, 1>("pure MFMA, 1-wave WGs x8/CU", 8);
  return 0;
}
