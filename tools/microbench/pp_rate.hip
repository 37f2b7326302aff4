// Ping-pong skeleton: 8 waves, groups of 4 staggered by one barrier; phase = 32 MFMAs (4 accumulators) | 12 ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: MFMA + barriers only; 1: + LDS reads feeding the MFMAs; 2: no barriers (free running); 3: free running + pipelined LDS reads
__global__ __launch_bounds__(512, 1) void k(float* out, const float* in, int stages) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 16384; i += 512) sm[i] = in[i & 1023];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float4 fa[4][2], fb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { fa[j][0] = fa[j][1] = fb[j] = make_float4(in[lane], in[lane + 64], in[lane + 128], in[lane + 192]); }
  if (MODE == 3) {
    for (int s = 0; s < stages; ++s) {
      const char* base = reinterpret_cast<const char*>(sm) + ((s & 3) * 8192) + lane * 16;
      float4 ga[2][2], gb[2];
      ga[0][0] = *reinterpret_cast<const float4*>(base);
      ga[0][1] = *reinterpret_cast<const float4*>(base + 4096);
      gb[0] = *reinterpret_cast<const float4*>(base + 32768);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < 3) {
          ga[(j + 1) & 1][0] = *reinterpret_cast<const float4*>(base + (j + 1) * 1024);
          ga[(j + 1) & 1][1] = *reinterpret_cast<const float4*>(base + (j + 1) * 1024 + 4096);
          gb[(j + 1) & 1] = *reinterpret_cast<const float4*>(base + (j + 1) * 1024 + 32768);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x16& a = (j & 1) ? acc[2 + i] : acc[i];
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j & 1][i].x, gb[j & 1].x, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j & 1][i].y, gb[j & 1].y, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j & 1][i].z, gb[j & 1].z, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j & 1][i].w, gb[j & 1].w, a, 0, 0, 0);
        }
      }
    }
  } else {
  if (MODE != 2 && wave >= 4) __builtin_amdgcn_s_barrier();
  for (int s = 0; s < stages; ++s) {
    if (MODE == 1) {
      const char* base = reinterpret_cast<const char*>(sm) + ((s & 3) * 8192) + lane * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fa[j][0] = *reinterpret_cast<const float4*>(base + j * 1024);
        fa[j][1] = *reinterpret_cast<const float4*>(base + j * 1024 + 4096);
        fb[j] = *reinterpret_cast<const float4*>(base + j * 1024 + 32768);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE != 2) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x16& a = (j & 1) ? acc[2 + i] : acc[i];
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][i].x, fb[j].x, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][i].y, fb[j].y, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][i].z, fb[j].z, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][i].w, fb[j].w, a, 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 2) __builtin_amdgcn_s_barrier();
  }
  if (MODE != 2 && wave < 4) __builtin_amdgcn_s_barrier();
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[i][e];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}
template <int MODE>
void run(int stages, float* out, float* in) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, out, in, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, out, in, stages);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 256.0 * 8 * stages * 32.0 * (32.0 * 32 * 2 * 2);
  printf("mode %d: %.2f ms  %.1f TFLOP/s\n", MODE, ms, flops / ms / 1e9);
}
int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 1024 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(4000, out, in); run<1>(4000, out, in); run<2>(4000, out, in); run<3>(4000, out, in);
  return 0;
}
