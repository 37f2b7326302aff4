// 3x3 / stride-1 / pad-1 fused convolution for gfx950 with 16-bit operands -- the MFMA-bound layers of
// CSPDarknet53 / PAN (darknetcsp.py:38-64 bottleneck 3x3, yolo_neck_csp.py out convs) and their data gradients.
//
// Why a second kernel.  The generic implicit-GEMM tiles (conv_mfma_h16.hip) fill LDS with one im2col slice per
// (tap, 64-channel chunk): per 64-deep K slice a 128x128 tile moves 32 KB L2 -> LDS for 2 MFLOP = 64 FLOP/B.  The
// matrix core retires 4069 FLOP/clk/CU, the L2 -> LDS path delivers ~70 GB/s per CU (~17 TB/s per chip): at 64 FLOP/B
// that path, not the MFMA, bounds the kernel near 1.1 PFLOP/s in theory and 0.65-0.85 measured.  Two things raise the
// intensity here:
//   * the three kw taps of one (chunk, kh) read the SAME input pixels shifted by one: one LDS image of BM + 2
//     consecutive source pixels serves all three (fragment row = output row + kw), so the activation operand is
//     fetched 3 instead of 9 times per chunk; what a shifted row must not see (left / right image border, the rows
//     above / below the image, the neighbouring image) is masked per lane by redirecting the fragment read to a
//     zero row -- one v_cndmask per fragment, no branch;
//   * the tile is 256 pixels x 128 channels on 8 waves (2 per SIMD, 64x64 outputs each): the weight operand is
//     fetched once per 256 pixels.
// Per (chunk, kh, kw) stage: 16 MFMAs per wave, 28 KB of LDS fill (12 KB activations amortised + 16 KB weights) for
// 4.2 MFLOP = 150 FLOP/B.  Ring: activations double-buffered per (chunk, kh) (2 x 40 KB), weights 4 slots of 16 KB,
// three stages of DMA in flight behind the one being computed; one workgroup per CU (144 KB LDS), one barrier per
// stage, hand-counted vmcnt (2*PB + 5 or 2*PB newer loads allowed, see the loop).
//
// Same epilogue, statistics and argument block as conv_mfma_h16.hip (conv_h16_common.h).
#include "conv_h16_common.h"

namespace yv4 {

constexpr int kC3Threads = 512;
constexpr int kC3BM = 256;
constexpr int kC3ARows = 320;       // BM + 2 source pixels, padded to 5 DMA passes of 64 rows; rows >= 258 stay zero
constexpr int kC3PA = kC3ARows / 64;
constexpr int kC3ZeroRow = 304;     // any row in [258, 320): never written with data
constexpr int kC3NB = 4;            // weight ring slots

template <bool BF16, int BN, bool PP>
__global__ __launch_bounds__(kC3Threads, 2) void conv3x3_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes) {
  typedef typename Elem<BF16>::V8 V8;
  static_assert(BN == 128 || BN == 64, "BN is 128 or 64");
  constexpr int TM = 2;
  constexpr int TN = BN / 64;              // wave tile 64 x (BN / 2)
  constexpr int PB = BN / 64;              // weight DMA passes of 64 rows
  constexpr int kRowB = 128;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_c3[];
  char* As = smem_c3;                                  // [2][kC3ARows][128 B]
  char* Bs = smem_c3 + 2 * kC3ARows * kRowB;           // [kC3NB][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;                // 0..3
  const int wn = wave & 1;
  const int r = lane & 31;
  const int h = lane >> 5;

  const unsigned nwg = gridDim.x;
  const unsigned bid = blockIdx.x;
  const unsigned xcd = bid & 7u, q8 = nwg >> 3, rem8 = nwg & 7u;
  const unsigned tile = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + (bid >> 3);
  const int tile_n = tile % p.tiles_n;
  const int tile_m = tile / p.tiles_n;
  const int m0 = tile_m * kC3BM;
  const int n0 = tile_n * BN;

  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc_h(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_c3;
  const int NHW = p.N * p.H * p.W;

  // ---- staging lanes: a DMA instruction of a wave fills 8 LDS rows (lane / 8) x 8 chunks (lane % 8) ----
  const int srow = 8 * wave + (lane >> 3);               // 0..63, + 64 per pass
  const int pc = lane & 7;
  const int lc = pc ^ ((srow >> 1) & 7);                 // (row >> 1) & 7 is the same for row + 64 q
  int a_s[kC3PA];                                        // source pixel of LDS row (srow + 64 q) for kh = 1
  unsigned a_off[kC3PA];                                 // its byte offset (wrapping arithmetic; used only when valid)
#pragma unroll
  for (int q = 0; q < kC3PA; ++q) {
    const int row = srow + 64 * q;
    a_s[q] = row < kC3BM + 2 ? m0 - 1 + row : (int)0x40000000;     // beyond the image for every kh: stays zero
    a_off[q] = (unsigned)((((int64_t)(m0 - 1 + row)) * p.x_cs + p.x_co + lc * 8) * 2);
  }
  unsigned b_off[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int co = n0 + srow + 64 * q;
    b_off[q] = co < p.Cout ? (unsigned)(((int64_t)co * p.Kw + lc * 8) * 2) : kOOB;
  }

  // ---- fragment read addresses ----
  unsigned a_rd[TM][3];
  unsigned mask9[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rr = wm * 64 + i * 32 + r;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = rr + kw;
      a_rd[i][kw] = (unsigned)(row * kRowB + ((((row >> 1) & 7) ^ h) << 4));
    }
    const int m = m0 + rr;
    unsigned mk = 0u;
    if (m < p.M) {
      const int hw = p.H * p.W;
      const int n = fd_div(m, p.fd_hw);
      const int rm = m - n * hw;
      const int ho = fd_div(rm, p.fd_wo);
      const int wo = rm - ho * p.W;
      mk = (unsigned)tap_mask(ho - 1, wo - 1, 3, 3, p.H, p.W);
    }
    mask9[i] = mk;
  }
  const unsigned zero_rd = (unsigned)(kC3ZeroRow * kRowB);     // all chunks of that row are zero: no swizzle needed
  unsigned b_rd[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * (BN / 2) + i * 32 + r;
    b_rd[i] = (unsigned)(row * kRowB + ((((row >> 1) & 7) ^ h) << 4));
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // the per-channel affine of the wave's column groups is requested BEFORE the K loop: fetched after it, its memory
  // latency (two dependent round trips per tile) sat exposed in front of the epilogue of a kernel that runs one
  // workgroup per CU
  const bool has2 = p.s2 != nullptr;
  const int ymask = p.out_f32 ? 3 : 7;
  const bool vec_ok = ((p.y_cs | p.y_co) & ymask) == 0 && (p.res == nullptr || ((p.r_cs | p.r_co) & 7) == 0);
  AffH af[TN];
  bool full[TN];
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int co = n0 + wn * (BN / 2) + jn * 32 + (lane & 3) * 8;
    full[jn] = vec_ok && co + 7 < p.Cout;
    if (full[jn]) load_affine_h(p, co, has2, af[jn]);
  }

  const int nchunks = p.Cin >> 6;
  const int G = YV4_ABLATE(p.ablate, 32) ? 0 : 3 * nchunks;   // (chunk, kh) groups; stage t = 3 g + kw  (ablate 32: measurement only)
  // issue-side walk: next stage to issue is (i_c0, i_kh, i_kw), its group index i_g
  int i_c0 = 0, i_kh = 0, i_kw = 0, i_g = 0, i_t = 0;

  // ISSUE: weights of stage i_t into ring slot i_t & 3 and, when the stage opens a group, the group's activation rows
  // into activation buffer i_g & 1.  Beyond the last stage the same number of (out-of-range, zero-filling) DMAs is
  // issued so that the vmcnt bookkeeping of the loop stays uniform.
#define YV4_C3_ISSUE()                                                                              \
  {                                                                                                 \
    const bool live = i_g < G;                                                                      \
    const unsigned lb_ = lds_base + (unsigned)((2 * kC3ARows + (i_t & 3) * BN + 8 * wave) * kRowB);  \
    const unsigned kb = (unsigned)((((i_kh * 3 + i_kw) * p.Cin) + i_c0) * 2);                        \
    _Pragma("unroll") for (int q = 0; q < PB; ++q)                                                  \
        lds_dma16_h(rsB, lb_ + 64 * q * kRowB, live ? b_off[q] : kOOB, live ? kb : 0u);              \
    if (i_kw == 0) {                                                                                \
      const unsigned la_ = lds_base + (unsigned)(((i_g & 1) * kC3ARows + 8 * wave) * kRowB);         \
      const int ds = (i_kh - 1) * p.W;                                                              \
      const unsigned step = (unsigned)(((int64_t)ds * p.x_cs + i_c0) * 2);                          \
      _Pragma("unroll") for (int q = 0; q < kC3PA; ++q) {                                           \
        const bool ok = live && (unsigned)(a_s[q] + ds) < (unsigned)NHW;                            \
        lds_dma16_h(rsA, la_ + 64 * q * kRowB, ok ? a_off[q] + step : kOOB, 0u);                     \
      }                                                                                             \
    }                                                                                               \
    i_t += 1;                                                                                       \
    i_kw += 1;                                                                                      \
    if (i_kw == 3) {                                                                                \
      i_kw = 0;                                                                                     \
      i_g += 1;                                                                                     \
      i_kh += 1;                                                                                    \
      if (i_kh == 3) { i_kh = 0; i_c0 += kHBK; }                                                    \
    }                                                                                               \
  }

// fragments of MFMA step j+1 are requested before the MFMAs of step j issue (two register sets): the LDS latency of a
// step hides behind the 4 x 32 matrix-pipe cycles of the previous one instead of in front of every group of four
#define YV4_C3_COMPUTE(KW, ABUF, BSLOT, MK)                                                         \
  {                                                                                                 \
    const char* as_ = As + (ABUF) * (kC3ARows * kRowB);                                             \
    const char* bs_ = Bs + (BSLOT) * (BN * kRowB);                                                  \
    unsigned ar[TM];                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
        ar[i] = ((MK[i] >> (KW)) & 1u) ? a_rd[i][KW] : zero_rd;                                      \
    V8 fa[2][TM], fb[2][TN];                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const V8*>(as_ + ar[i]);      \
    _Pragma("unroll") for (int i = 0; i < TN; ++i) fb[0][i] = *reinterpret_cast<const V8*>(bs_ + b_rd[i]);    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
      if (j < 3) {                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
            fa[(j + 1) & 1][i] = *reinterpret_cast<const V8*>(as_ + (ar[i] ^ (unsigned)((j + 1) << 5)));   \
        _Pragma("unroll") for (int i = 0; i < TN; ++i)                                              \
            fb[(j + 1) & 1][i] = *reinterpret_cast<const V8*>(bs_ + (b_rd[i] ^ (unsigned)((j + 1) << 5))); \
      }                                                                                             \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
        _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                           \
            acc[i][jn] = Elem<BF16>::mfma(fa[j & 1][i], fb[j & 1][jn], acc[i][jn]);                 \
    }                                                                                               \
  }

#define YV4_C3_WAIT(NEWER) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEWER) : "memory")

  // ---- ping-pong form (PP): the 16 fragments of a stage are read into registers in a LOAD phase, the 16 MFMAs run in
  // the next phase, two barriers per stage, and waves 4-7 run one phase behind waves 0-3 (one extra barrier up front):
  // on every SIMD one wave is in its MFMA phase while its partner reads LDS and issues DMA (MI355X_MICROARCH.md "Two
  // waves per SIMD"; the 256^2 8-phase GEMM template of cdna_hip_programming.md is built the same way).  Reads of a
  // stage are retired (lgkmcnt(0)) BEFORE the barrier that ends their phase, so the DMA the other group issues one
  // phase later into the same ring slot cannot overtake them.
  V8 fa4[4][TM], fb4[4][TN];
#define YV4_C3_LOAD(KW, ABUF, BSLOT, MK)                                                            \
  {                                                                                                 \
    const char* as_ = As + (ABUF) * (kC3ARows * kRowB);                                             \
    const char* bs_ = Bs + (BSLOT) * (BN * kRowB);                                                  \
    unsigned ar[TM];                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
        ar[i] = ((MK[i] >> (KW)) & 1u) ? a_rd[i][KW] : zero_rd;                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
          fa4[j][i] = *reinterpret_cast<const V8*>(as_ + (ar[i] ^ (unsigned)(j << 5)));             \
      _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                \
          fb4[j][i] = *reinterpret_cast<const V8*>(bs_ + (b_rd[i] ^ (unsigned)(j << 5)));           \
    }                                                                                               \
  }
#define YV4_C3_MFMA()                                                                               \
  {                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
        _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                           \
            acc[i][jn] = Elem<BF16>::mfma(fa4[j][i], fb4[j][jn], acc[i][jn]);                       \
  }
#define YV4_C3_PP_STAGE(KW, NEWER)                                                                  \
  {                                                                                                 \
    YV4_C3_ISSUE();                                                                                 \
    YV4_C3_LOAD(KW, ab, (t0 + (KW)) & 3, mk3);                                                      \
    YV4_C3_WAIT(NEWER);                                                                             \
    __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    YV4_C3_MFMA();                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_barrier();                                                                   \
  }

  // prologue: stages 0, 1, 2 in flight, then wait for stage 0 (weights 0 + activation group 0; newer: weights 1, 2)
  YV4_C3_ISSUE();
  YV4_C3_ISSUE();
  YV4_C3_ISSUE();
  YV4_C3_WAIT(2 * PB);
  __builtin_amdgcn_s_barrier();

  if (PP) {
    if (wm >= 2) __builtin_amdgcn_s_barrier();         // waves 4-7 run one phase behind waves 0-3
    for (int g = 0; g < G; ++g) {
      const int kh = g - 3 * (g / 3);
      unsigned mk3[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) mk3[i] = mask9[i] >> (3 * kh);
      const int ab = g & 1;
      const int t0 = 3 * g;
      YV4_C3_PP_STAGE(0, 2 * PB + kC3PA);
      YV4_C3_PP_STAGE(1, 2 * PB + kC3PA);
      YV4_C3_PP_STAGE(2, 2 * PB);
    }
    if (wm < 2) __builtin_amdgcn_s_barrier();          // same number of barriers for both halves
  } else
  for (int g = 0; g < G; ++g) {
    const int kh = g - 3 * (g / 3);
    unsigned mk3[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) mk3[i] = mask9[i] >> (3 * kh);
    const int ab = g & 1;
    const int t0 = 3 * g;
    // kw = 0: issue stage t+3 (weights + the NEXT group's activations), compute, wait for stage t+1:
    //         newer than what it needs = weights t+2, weights t+3, activations g+1
    if (!YV4_ABLATE(p.ablate, 1)) YV4_C3_ISSUE();
    __builtin_amdgcn_s_setprio(1);
    if (!YV4_ABLATE(p.ablate, 2)) YV4_C3_COMPUTE(0, ab, (t0 + 0) & 3, mk3);
    __builtin_amdgcn_s_setprio(0);
    if (!YV4_ABLATE(p.ablate, 8)) YV4_C3_WAIT(2 * PB + kC3PA);
    if (!YV4_ABLATE(p.ablate, 4)) __builtin_amdgcn_s_barrier();
    // kw = 1: newer = weights t+2 + activations g+1 (issued at kw = 0 ... no: issued with stage t+2 = 3g+3 at kw = 0 of
    //         this group), weights t+3
    if (!YV4_ABLATE(p.ablate, 1)) YV4_C3_ISSUE();
    __builtin_amdgcn_s_setprio(1);
    if (!YV4_ABLATE(p.ablate, 2)) YV4_C3_COMPUTE(1, ab, (t0 + 1) & 3, mk3);
    __builtin_amdgcn_s_setprio(0);
    if (!YV4_ABLATE(p.ablate, 8)) YV4_C3_WAIT(2 * PB + kC3PA);
    if (!YV4_ABLATE(p.ablate, 4)) __builtin_amdgcn_s_barrier();
    // kw = 2: the next stage opens group g+1 and needs its activations: newer = weights t+2, weights t+3 only
    if (!YV4_ABLATE(p.ablate, 1)) YV4_C3_ISSUE();
    __builtin_amdgcn_s_setprio(1);
    if (!YV4_ABLATE(p.ablate, 2)) YV4_C3_COMPUTE(2, ab, (t0 + 2) & 3, mk3);
    __builtin_amdgcn_s_setprio(0);
    if (!YV4_ABLATE(p.ablate, 8)) YV4_C3_WAIT(2 * PB);
    if (!YV4_ABLATE(p.ablate, 4)) __builtin_amdgcn_s_barrier();
  }
#undef YV4_C3_PP_STAGE
#undef YV4_C3_MFMA
#undef YV4_C3_LOAD
#undef YV4_C3_ISSUE
#undef YV4_C3_COMPUTE
#undef YV4_C3_WAIT
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the zero-filling tail DMAs
  __builtin_amdgcn_s_barrier();

  // ---- BatchNorm statistics of the tile (training, identity epilogue), as conv_mfma_h16.hip ----
  if (p.stats) {
    typedef typename Elem<BF16>::T TS;
    double* rep = p.stats + (size_t)(tile_m & (YV4_STATS_REPLICAS - 1)) * 2 * p.Cout;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float su = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * 64 + i * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          const float v = p.out_f32 ? acc[i][jn][e] : (float)(TS)acc[i][jn][e];
          if (m < p.M) { su += v; sq += v * v; }
        }
      }
      su += __shfl_xor(su, 32);
      sq += __shfl_xor(sq, 32);
      const int col = n0 + wn * (BN / 2) + jn * 32 + r;
      if (h == 0 && col < p.Cout) {
        atomicAdd(&rep[col], (double)su);
        atomicAdd(&rep[p.Cout + col], (double)sq);
      }
    }
  }
  if (YV4_ABLATE(p.ablate, 64)) return;                 // measurement only: no epilogue at all
  // every accumulator tile of the wave gets its own 32 x 36 fp32 patch (8 waves x TM*TN x 4608 B <= the K-loop carve):
  // stage all of them, then finish all of them
  float* ep = reinterpret_cast<float*>(smem_c3) + wave * (TM * TN * 32 * 36);
  static_assert((size_t)8 * TM * TN * 32 * 36 * 4 <= (size_t)(2 * kC3ARows + kC3NB * BN) * 128, "epilogue patches fit");
  const AffH none{};
#pragma unroll
  for (int jn = 0; jn < TN; ++jn)
#pragma unroll
    for (int i = 0; i < TM; ++i)
      epilogue_tile_h<BF16, true, false>(p, acc[i][jn], ep + (i * TN + jn) * (32 * 36), lane, 0, 0, false, has2, none);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int cb = n0 + wn * (BN / 2) + jn * 32;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      epilogue_tile_h<BF16, false, true>(p, acc[i][jn], ep + (i * TN + jn) * (32 * 36), lane, m0 + wm * 64 + i * 32, cb,
                                         full[jn], has2, af[jn]);
  }
}

template <bool BF16, int BN, bool PP>
static int launch_c3(const ConvArgsH& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(2 * kC3ARows + kC3NB * BN) * 128;
  ConvArgsH p = a;
  const int tiles_m = (p.M + kC3BM - 1) / kC3BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  p.fd_hw = make_fastdiv((unsigned)(p.H * p.W));
  p.fd_wo = make_fastdiv((unsigned)p.W);
  const long long tiles = (long long)tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv3x3 h16: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  const long long xb = (long long)p.N * p.H * p.W * p.x_cs * 2, wb = (long long)p.Cout * p.Kw * 2;
  auto kern = conv3x3_h16_kernel<BF16, BN, PP>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), lds, "conv3x3_h16")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(kC3Threads), lds, stream, p, (unsigned)xb, (unsigned)wb);
  YV4_CHECK_LAUNCH("conv3x3_h16");
  return YV4_OK;
}

// Is this layer in the kernel's domain?  3x3, stride 1, pad 1 (so Ho = H, Wo = W), 64-channel chunks of input.
bool conv3x3_h16_applies(const ConvArgsH& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && (a.Cin & 63) == 0 &&
         !a.ys_on && a.Cout >= 64;
}

int conv3x3_h16_launch(const ConvArgsH& a, bool bf16, int tile, hipStream_t s) {
  const bool wide = tile == YV4_HTILE_C3_256x128 || (tile != YV4_HTILE_C3_256x64 && a.Cout > 64);
  // YV4_C3_PP=0: the lock-step form (all eight waves in the same phase), kept for A/B measurement
  static const bool pp = YV4_ENV_INT("YV4_C3_PP", 1) != 0;
  if (pp) {
    if (bf16) return wide ? launch_c3<true, 128, true>(a, s) : launch_c3<true, 64, true>(a, s);
    return wide ? launch_c3<false, 128, true>(a, s) : launch_c3<false, 64, true>(a, s);
  }
  if (bf16) return wide ? launch_c3<true, 128, false>(a, s) : launch_c3<true, 64, false>(a, s);
  return wide ? launch_c3<false, 128, false>(a, s) : launch_c3<false, 64, false>(a, s);
}

}  // namespace yv4
