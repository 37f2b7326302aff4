// Skeleton of the fp32 wide 3x3 kernel's K loop (conv3x3_wide_f32.hip, six pixel tiles per wave): 8 waves, a K tile = four phases
// of 48 v_mfma_f32_16x16x4_f32 on 24 accumulator tiles, 10 / 4 / 6 / 0 ds_read_b128 in front of the phases, one barrier per K
// tile, four 1-KB LDS-DMA pieces per wave and K tile.  Which part costs the loop its 8.5 %?
//   mode 0  MFMAs only, free running
//   mode 1  + the fragment reads in front of each phase
//   mode 2  + one workgroup barrier per K tile
//   mode 3  + four LDS-DMA pieces per wave at the top of the K tile, s_waitcnt vmcnt(0) in front of the barrier  (= the kernel)
//   mode 4  mode 3 with the pieces spread: one in front of each phase
//   mode 5  mode 3 with waves 4-7 half a K tile behind (two barriers per K tile: waves 0-3 run P1 P2 | P3 P4, waves 4-7 P3 P4 | P1 P2)
//   mode 6  mode 2 with waves 4-7 half a K tile behind (no DMA)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/w3f_skeleton.hip -o /tmp/w3f_skeleton && /tmp/w3f_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* in, const float* big, unsigned big_bytes, int ktiles) {
  extern __shared__ __attribute__((aligned(16))) char sm[];   // 128 KB: [0, 64 K) "image", [64 K, 128 K) two weight slots
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  for (int i = threadIdx.x; i < 32768; i += 512) reinterpret_cast<float*>(sm)[i] = in[i & 1023];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)sm;
  u32x4 rs;
  {
    const unsigned long long a = reinterpret_cast<unsigned long long>(big);
    rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    rs.z = __builtin_amdgcn_readfirstlane(big_bytes);
    rs.w = 0x00020000u;
  }
  f32x4 acc[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 wf[4][2], pf[3][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wf[t][ks] = f32x4{in[lane], in[lane + 64], in[lane + 128], in[lane + 192]};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) pf[i][ks] = f32x4{in[lane + 256], in[lane + 320], in[lane + 384], in[lane + 448]};
  const unsigned a_rd = (unsigned)(wave * 6144 + lane * 16);                 // conflict-free: consecutive lanes, consecutive chunks
  const unsigned w_rd = (unsigned)(65536 + (wave & 3) * 8192 + lane * 16);
  const unsigned voff = (unsigned)((blockIdx.x * 512 + threadIdx.x) * 16) % (big_bytes - 65536);

#define READ_W(T0, T1, SLOT)                                                                                            \
  if (MODE >= 1) {                                                                                                      \
    _Pragma("unroll") for (int t = T0; t < T1; ++t) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                    \
        wf[t][ks] = *reinterpret_cast<const f32x4*>(sm + w_rd + (SLOT) * 32768 + t * 1024 + ks * 4096);                 \
  }
#define READ_P(H)                                                                                                       \
  if (MODE >= 1) {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                      \
        pf[i][ks] = *reinterpret_cast<const f32x4*>(sm + a_rd + ((H) * 3 + i) * 1024 + ks * 32768 % 49152);             \
  }
#define PHASE(T0, H)                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);                                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 4; ++j)                        \
      _Pragma("unroll") for (int t = T0; t < T0 + 2; ++t) _Pragma("unroll") for (int i = 0; i < 3; ++i)                 \
          acc[(H) * 3 + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][ks][j], pf[i][ks][j], acc[(H) * 3 + i][t], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                                        \
  __builtin_amdgcn_sched_barrier(0);
#define PIECES(Q0, Q1, SLOT, KT)                                                                                        \
  _Pragma("unroll") for (int q = Q0; q < Q1; ++q)                                                                       \
      lds_dma16(rs, lds_base + 65536u + (SLOT) * 32768u + (unsigned)((wave * 4 + q) * 1024), voff, (unsigned)(((KT) * 4 + q) * 4096 & 0xffff));

  constexpr bool DMA = MODE == 3 || MODE == 4 || MODE == 5;
  constexpr bool STAG = MODE == 5 || MODE == 6;
  if (!STAG) {
    for (int kt = 0; kt < ktiles; ++kt) {
      const unsigned slot = kt & 1u;
      if (MODE == 3) PIECES(0, 4, slot ^ 1u, kt)
      if (MODE == 4) PIECES(0, 1, slot ^ 1u, kt)
      READ_W(0, 2, slot) READ_P(0)
      PHASE(0, 0)
      if (MODE == 4) PIECES(1, 2, slot ^ 1u, kt)
      READ_W(2, 4, slot)
      PHASE(2, 0)
      if (MODE == 4) PIECES(2, 3, slot ^ 1u, kt)
      READ_P(1)
      PHASE(2, 1)
      if (MODE == 4) PIECES(3, 4, slot ^ 1u, kt)
      PHASE(0, 1)
      if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (MODE >= 2) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  } else {
    // waves 0-3: [P1 P2] bar [P3 P4] bar ...; waves 4-7 start with one extra barrier and run [P3 P4](t-1) | [P1 P2](t)
    const bool grpB = wave >= 4;
    for (int h = 0; h < 2 * ktiles; ++h) {
      const int kt = (h + (grpB ? 1 : 0)) >> 1;       // waves 4-7: half h works on K tile (h + 1) / 2's first half when h is odd
      const bool firsthalf = ((h & 1) == 0) != grpB;
      const unsigned slot = kt & 1u;
      if (firsthalf) {
        // the weights of K tile kt + 1 go to the slot K tile kt - 1 used: free once BOTH groups are past it, i.e. from the half in
        // which waves 4-7 start K tile kt -- issued by waves 4-7 there (waves 0-3 are in their second half, deep in MFMAs)
        if (DMA && grpB) PIECES(0, 4, slot ^ 1u, kt)
        if (DMA && grpB) { const int w2 = wave - 4; (void)w2; }
        READ_W(0, 2, slot) READ_P(0)
        PHASE(0, 0)
        READ_W(2, 4, slot)
        PHASE(2, 0)
      } else {
        if (DMA && !grpB) PIECES(0, 4, slot ^ 1u, kt)
        READ_P(1)
        PHASE(2, 1)
        PHASE(0, 1)
      }
      if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  float tsum = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) tsum += acc[i][t][0] + acc[i][t][1] + acc[i][t][2] + acc[i][t][3];
  out[blockIdx.x * 512 + threadIdx.x] = tsum;
}

template <int MODE>
void run(int ktiles, float* out, float* in, float* big, unsigned big_bytes) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 131072, 0, out, in, big, big_bytes, ktiles);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 131072, 0, out, in, big, big_bytes, ktiles);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 8 * ktiles * 192.0 * (16.0 * 16 * 4 * 2);
  printf("mode %d: %.3f ms  %.1f TFLOP/s = %.3f of 157.3\n", MODE, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}

int main() {
  float *out, *in, *big;
  const unsigned big_bytes = 64u << 20;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&big, big_bytes);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMemset(big, 0, big_bytes);
  const int kt = 720;   // ten layers' worth: ~4 ms
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(kt, out, in, big, big_bytes); run<1>(kt, out, in, big, big_bytes); run<2>(kt, out, in, big, big_bytes);
    run<3>(kt, out, in, big, big_bytes); run<4>(kt, out, in, big, big_bytes); run<5>(kt, out, in, big, big_bytes);
    run<6>(kt, out, in, big, big_bytes);
  }
  return 0;
}
