// 3x3 / stride-1 / pad-1 fused convolution for gfx950 with 16-bit operands: the MFMA-bound layers of CSPDarknet53 / PAN
// (darknetcsp.py:38-64 bottleneck 3x3, yolo_neck_csp.py out convs) and their data gradients -- PERSISTENT ping-pong form.
//
// Geometry (unchanged from round 2's conv3x3_h16.hip, which this file replaces): one 8-wave workgroup per CU owns
// 256 pixels x 128 channels; a stage is one (64-channel chunk, kh, kw) tap = 16 MFMAs (32x32x16) per wave; the three kw
// taps of a (chunk, kh) read ONE LDS image of the 258 source pixels (fragment row = output row + kw; what a shifted row
// must not see -- image borders, the neighbouring image -- is masked by redirecting the lane's read to a zero row), so
// the activations are fetched 3 instead of 9 times per chunk: 28 KB of LDS fill per 4.2 MFLOP stage, under what the
// L2 -> LDS path delivers (the generic tiles of conv_mfma_h16.hip move 24 KB per 1.05 MFLOP and sit at that limit).
// Waves 4-7 run one phase behind waves 0-3: on every SIMD one wave is in its MFMA phase while its partner reads LDS,
// issues DMA and waits (MI355X_MICROARCH.md "Two waves per SIMD").
//
// What round 2 measured on that kernel: 25 of a layer's 71 us were outside the K loop -- per tile 3 us of prologue
// (address set-up, first-slice latency) and 6 us of epilogue, on one workgroup per CU with nothing to overlap them.
// What changed here:
//   * the workgroup is persistent (grid = min(tiles, CUs)) and the LDS-DMA ring runs ON ACROSS TILES: the issue side
//     walks (tile, chunk, kh, kw) three stages ahead of the compute side with its own per-tile lane constants, so the
//     first stages of tile t + 1 land while tile t is in its last stages and its epilogue -- no per-tile prologue;
//   * the epilogue needs no LDS and no barrier: affine + activation in the MFMA's C layout (a lane owns ONE output
//     channel of 16 rows), then lanes (r even, r odd) trade halves through one DPP swap per value, after which a lane
//     owns channel PAIRS of 8 rows: residual add, second affine + activation, and dword stores of two 16-bit values
//     (64 contiguous bytes per 16 lanes; these layers move 0.4 bytes per kFLOP, store shape is not their bound).  The
//     round-2 form staged every accumulator tile through LDS patches carved from the K-loop buffers, which forced a
//     drain + barrier before and after, and kept 64 VGPRs of per-lane affine alive across the K loop (68 spilled);
//   * same K order as the generic tiles (chunk-major, taps inside, four 16-deep MFMA steps per tap) and the same
//     epilogue expressions on the same fp32 accumulators, so a layer gives the same bits whichever kernel a batch size
//     selects (tests/test_gpu_h16.py::test_pp3x3_matches_generic_bitwise).
//
// Phases and vmcnt bookkeeping.  A wave alternates LOAD(t) = { the PB weight pieces of stage t + 3; 16 fragment reads of
// stage t; wait; barrier } and MFMA(t) = { 16 MFMAs, with the PA activation pieces of stage t + 3 issued BETWEEN them when
// that stage opens a (chunk, kh) group, i.e. at kw = 0; barrier }.  The split is the measured balance
// (profiles/r03_pp3_ablation.md): every 1 KB piece costs its wave ~100 cycles of issue wherever it is placed, the reads
// ~500 and the 16 MFMAs ~720 cycles per stage, and an interval lasts as long as the longer of one group's LOAD and the
// other's MFMA phase.  In program order a wave has issued ... B(t+1) A(t+1) B(t+2) A(t+2) B(t+3) when it waits in LOAD(t);
// it needs its own B(t+1) and A(t+1) landed, so B(t+2), A(t+2), B(t+3) may stay in flight: 2 PB pieces at kw = 0 and 2,
// 2 PB + PA at kw = 1 (A(t+2) exists when stage t + 2 opens a group).  The epilogue's loads and stores enter the same
// in-order counter BETWEEN DMAs: they make a counted wait more conservative (it then also covers an older DMA), never
// less.  Reads of a staged buffer come one barrier after the wait that retires it; a buffer is re-filled at least one
// barrier after the lgkmcnt(0) that retired its last reads.
#include "conv_h16_common.h"

namespace yv4 {

constexpr int kP3Threads = 512;
constexpr int kP3BM = 256;
constexpr int kP3BN = 128;
constexpr int kP3ARows = 320;       // BM + 2 source pixels, padded to 5 DMA passes of 64 rows; rows >= 258 stay zero
constexpr int kP3PA = kP3ARows / 64;
constexpr int kP3PB = kP3BN / 64;
constexpr int kP3ZeroRow = 304;     // any row in [258, 320): only ever zero-filled
constexpr int kP3NB = 4;            // weight ring slots
constexpr int kP3RingBytes = (2 * kP3ARows + kP3NB * kP3BN) * 128;
constexpr int kP3MaxCout = 1024;     // the per-channel affine of the whole layer lives in the last 16 KB of LDS
constexpr int kP3Lds = kP3RingBytes + 4 * kP3MaxCout * 4;

template <bool BF16>
__global__ __launch_bounds__(kP3Threads, 2) void conv3x3_pp_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef typename Elem<BF16>::V8 V8;
  constexpr int TM = 2, TN = 2;            // wave tile 64 x 64
  constexpr int PA = kP3PA, PB = kP3PB;
  constexpr int kRowB = 128;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_p3[];
  char* As = smem_p3;                                  // [2][kP3ARows][128 B]
  char* Bs = smem_p3 + 2 * kP3ARows * kRowB;           // [kP3NB][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;                // 0..3: 64-row slab
  const int wn = wave & 1;                 // 64-column slab
  const int r = lane & 31;
  const int h = lane >> 5;

  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc_h(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_p3;
  const int NHW = p.N * p.H * p.W;
  const int nwg = (int)gridDim.x;

  // virtual tile index -> tile: each XCD (workgroups with equal id mod 8) walks a contiguous run of tiles, tile_n fastest
  const unsigned q8 = (unsigned)ntiles >> 3, rem8 = (unsigned)ntiles & 7u;
  auto tile_of = [&](int vt) -> unsigned {
    const unsigned x = (unsigned)vt & 7u;
    return (x < rem8 ? x * (q8 + 1) : rem8 * (q8 + 1) + (x - rem8) * q8) + ((unsigned)vt >> 3);
  };

  // ---- staging lanes: a DMA instruction of a wave fills 8 LDS rows (lane / 8) x 8 chunks (lane % 8) ----
  const int srow = 8 * wave + (lane >> 3);               // 0..63, + 64 per pass
  const int pc = lane & 7;
  const int lc = pc ^ ((srow >> 1) & 7);                 // (row >> 1) & 7 is the same for row + 64 q
  int a_s[PA];                                           // source pixel of LDS row (srow + 64 q) for kh = 1
  unsigned a_off[PA];                                    // its byte offset (wrapping arithmetic; used only when valid)
  unsigned b_off[PB];
  int i_vt = (int)blockIdx.x;                            // issue side: virtual tile, walk inside it, global counters
  auto issue_tile_setup = [&]() {
    const bool live = i_vt < ntiles;
    const unsigned tile = live ? tile_of(i_vt) : 0u;
    const int tn = (int)(tile % (unsigned)p.tiles_n);
    const int m0i = YV4_ABLATE(p.ablate, 32) ? 0 : (int)(tile / (unsigned)p.tiles_n) * kP3BM;   // (bit 32: every tile reads tile 0's pixels)
    const int n0i = tn * kP3BN;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int row = srow + 64 * q;
      a_s[q] = (live && row < kP3BM + 2) ? m0i - 1 + row : (int)0x40000000;     // beyond the image for every kh: zero
      a_off[q] = (unsigned)((((int64_t)(m0i - 1 + row)) * p.x_cs + p.x_co + lc * 8) * 2);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int co = n0i + srow + 64 * q;
      b_off[q] = (live && co < p.Cout) ? (unsigned)(((int64_t)co * p.Kw + lc * 8) * 2) : kOOB;
    }
  };
  issue_tile_setup();

  // ---- fragment read addresses (tile-independent) ----
  unsigned a_rd[TM][3];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rr = wm * 64 + i * 32 + r;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = rr + kw;
      a_rd[i][kw] = (unsigned)(row * kRowB + ((((row >> 1) & 7) ^ h) << 4));
    }
  }
  const unsigned zero_rd = (unsigned)(kP3ZeroRow * kRowB);     // all chunks of that row are zero: no swizzle needed
  unsigned b_rd[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * 64 + i * 32 + r;
    b_rd[i] = (unsigned)(row * kRowB + ((((row >> 1) & 7) ^ h) << 4));
  }

  const int nchunks = p.Cin >> 6;
  const int G = 3 * nchunks;               // (chunk, kh) groups per tile; stage = 3 g + kw
  int i_c0 = 0, i_kh = 0, i_g = 0;
  unsigned i_t = 0u, i_gg = 0u;            // global stage / group counters: ring slot i_t & 3, activation buffer i_gg & 1

  // ---- the layer's affine into LDS, once per workgroup (the epilogue of every tile reads it from there) ----
  float* aff = reinterpret_cast<float*>(smem_p3 + kP3RingBytes);
  const bool has2 = p.s2 != nullptr;
  for (int c = tid; c < p.Cout; c += kP3Threads) {
    aff[c] = p.s1[c];
    aff[p.Cout + c] = p.t1[c];
    aff[2 * p.Cout + c] = has2 ? p.s2[c] : 1.f;
    aff[3 * p.Cout + c] = has2 ? p.t2[c] : 0.f;
  }

  // ISSUE of the stage three ahead, in three pieces that the MFMA phase interleaves with its MFMAs: the weight pieces
  // into ring slot i_t & 3; when the stage opens a group (its kw = the current stage's KW = 0) the group's activation
  // rows into activation buffer i_gg & 1; then the walk.  Beyond the last tile the same number of (out-of-range,
  // zero-filling) DMAs is issued so that the vmcnt bookkeeping stays uniform.
#define YV4_P3_ISSUE_B(KW)                                                                          \
  {                                                                                                 \
    const unsigned lb_ = lds_base + (unsigned)((2 * kP3ARows + (int)(i_t & 3u) * kP3BN + 8 * wave) * kRowB); \
    const unsigned kb = (unsigned)((((i_kh * 3 + (KW)) * p.Cin) + i_c0) * 2);                        \
    _Pragma("unroll") for (int q = 0; q < PB; ++q)                                                  \
        lds_dma16_h(rsB, lb_ + 64 * q * kRowB, b_off[q], kb);   /* (the range check sees voffset only) */ \
  }
#define YV4_P3_ISSUE_A(Q0, Q1)                                                                      \
  {                                                                                                 \
    const unsigned la_ = lds_base + (unsigned)(((int)(i_gg & 1u) * kP3ARows + 8 * wave) * kRowB);    \
    const int ds = (i_kh - 1) * p.W;                                                                \
    const unsigned step = (unsigned)(((int64_t)ds * p.x_cs + i_c0) * 2);                            \
    _Pragma("unroll") for (int q = (Q0); q < (Q1); ++q) {                                           \
      const bool ok = (unsigned)(a_s[q] + ds) < (unsigned)NHW;                                      \
      lds_dma16_h(rsA, la_ + 64 * q * kRowB, ok ? a_off[q] + step : kOOB, 0u);                       \
    }                                                                                               \
  }
#define YV4_P3_ADVANCE(KW)                                                                          \
  {                                                                                                 \
    i_t += 1u;                                                                                      \
    if ((KW) == 2) {                                                                                \
      i_gg += 1u;                                                                                   \
      i_g += 1;                                                                                     \
      i_kh += 1;                                                                                    \
      if (i_kh == 3) { i_kh = 0; i_c0 += kHBK; }                                                    \
      if (i_g == G) {                                                                               \
        i_g = 0; i_c0 = 0;                                                                          \
        i_vt += nwg;                                                                                \
        issue_tile_setup();                                                                         \
      }                                                                                             \
    }                                                                                               \
  }

#define YV4_P3_WAIT(NEWER) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEWER) : "memory")

  V8 fa4[4][TM] = {}, fb4[4][TN] = {};
  f32x16 acc[TM][TN];
#define YV4_P3_LOAD(KW, ABUF, BSLOT, MK)                                                            \
  {                                                                                                 \
    const char* as_ = As + (ABUF) * (kP3ARows * kRowB);                                             \
    const char* bs_ = Bs + (BSLOT) * (kP3BN * kRowB);                                               \
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
#define YV4_P3_MFMA_J(J)                                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
      _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                             \
          acc[i][jn] = Elem<BF16>::mfma(fa4[J][i], fb4[J][jn], acc[i][jn]);                         \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
// (measurement build only, YV4_H16_ABLATE: 1 no DMA after the prologue, 2 no MFMA, 4 no fragment reads, 8 no epilogue,
// 16 no vmcnt wait, 32 every tile reads tile 0's pixels, 64 print per-phase cycle sums, 128 no activation pieces, 256 no
// weight pieces -- wrong results on purpose, to time the loop without one of its parts)
#ifdef YV4_MEASURE
#define YV4_P3_STAMP(SLOT) if (YV4_ABLATE(p.ablate, 64)) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tsum[SLOT] += n_ - tlast; tlast = n_; }
#else
#define YV4_P3_STAMP(SLOT)
#endif
#define YV4_P3_STAGE(KW, NEWER)                                                                     \
  {                                                                                                 \
    YV4_P3_STAMP(5);                                                                                \
    if (!YV4_ABLATE(p.ablate, 1) && !YV4_ABLATE(p.ablate, 256)) YV4_P3_ISSUE_B(KW);                 \
    if (!YV4_ABLATE(p.ablate, 4)) YV4_P3_LOAD(KW, ab, (t0 + (KW)) & 3u, mk3);                       \
    YV4_P3_STAMP(0);                                                                                \
    if (!YV4_ABLATE(p.ablate, 16)) YV4_P3_WAIT(NEWER); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    YV4_P3_STAMP(1);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                   \
    asm volatile("" ::: "memory");                                                                  \
    YV4_P3_STAMP(2);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    if (!YV4_ABLATE(p.ablate, 2)) YV4_P3_MFMA_J(0);                                                 \
    if ((KW) == 0 && !YV4_ABLATE(p.ablate, 1) && !YV4_ABLATE(p.ablate, 128)) YV4_P3_ISSUE_A(0, 2);  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (!YV4_ABLATE(p.ablate, 2)) YV4_P3_MFMA_J(1);                                                 \
    if ((KW) == 0 && !YV4_ABLATE(p.ablate, 1) && !YV4_ABLATE(p.ablate, 128)) YV4_P3_ISSUE_A(2, 4);  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (!YV4_ABLATE(p.ablate, 2)) YV4_P3_MFMA_J(2);                                                 \
    if ((KW) == 0 && !YV4_ABLATE(p.ablate, 1) && !YV4_ABLATE(p.ablate, 128)) YV4_P3_ISSUE_A(4, PA); \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (!YV4_ABLATE(p.ablate, 2)) YV4_P3_MFMA_J(3);                                                 \
    YV4_P3_ADVANCE(KW);                                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    YV4_P3_STAMP(3);                                                                                \
    __builtin_amdgcn_s_barrier();                                                                   \
    asm volatile("" ::: "memory");   /* no LDS read of the next stage may move above the barrier */  \
    YV4_P3_STAMP(4);                                                                                \
  }

  // prologue: stages 0, 1, 2 of the first tile in flight, then wait for stage 0 (newer: weights 1, 2)
  YV4_P3_ISSUE_B(0); YV4_P3_ISSUE_A(0, PA); YV4_P3_ADVANCE(0);
  YV4_P3_ISSUE_B(1); YV4_P3_ADVANCE(1);
  YV4_P3_ISSUE_B(2); YV4_P3_ADVANCE(2);
  YV4_P3_WAIT(2 * PB);
  __builtin_amdgcn_s_barrier();                        // (also publishes the affine written above)
  if (wm >= 2) __builtin_amdgcn_s_barrier();           // waves 4-7 run one phase behind waves 0-3

  unsigned gg = 0u;                                    // compute side: global group counter
#ifdef YV4_MEASURE
  // bit 64: cycles per wave spent in [0] fragment-read issue, [1] vmcnt/lgkmcnt wait, [2] barrier after LOAD, [3] MFMA phase
  // (with the DMA pieces), [4] barrier after MFMA, [5] between stages (tile set-up, epilogue); printed by two workgroups
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tbegin = tlast;
#endif
  for (int vt = (int)blockIdx.x; vt < ntiles; vt += nwg) {
    const unsigned tile = tile_of(vt);
    const int tile_n = (int)(tile % (unsigned)p.tiles_n);
    const int tile_m = (int)(tile / (unsigned)p.tiles_n);
    const int m0 = tile_m * kP3BM;
    const int n0 = tile_n * kP3BN;
    unsigned mask9[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * 64 + i * 32 + r;
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
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int kh = 0;
    for (int g = 0; g < G; ++g) {
      unsigned mk3[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) mk3[i] = mask9[i] >> (3 * kh);
      kh = kh == 2 ? 0 : kh + 1;
      const unsigned ab = gg & 1u;
      const unsigned t0 = 3u * gg;
      gg += 1u;
      YV4_P3_STAGE(0, 2 * PB);
      YV4_P3_STAGE(1, 2 * PB + PA);
      YV4_P3_STAGE(2, 2 * PB);
    }

    // ---- BatchNorm statistics of the tile (training forward, identity epilogue), as conv_mfma_h16.hip ----
    if (p.stats) {
      typedef typename Elem<BF16>::T TS;
      const StatRep rep = stat_rep(p.stats, (unsigned)(tile_m), p.Cout);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mb = m0 + wm * 64 + i * 32 + 4 * h;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = mb + (e & 3) + 8 * (e >> 2);
            const float v = (float)(TS)acc[i][jn][e];
            if (m < p.M) { su += v; sq += v * v; }
          }
        }
        su += __shfl_xor(su, 32);
        sq += __shfl_xor(sq, 32);
        const int col = n0 + wn * 64 + jn * 32 + r;
        if (h == 0 && col < p.Cout) {
          stat_add(rep, col, su);
          stat_add(rep, p.Cout + col, sq);
        }
      }
    }
    // ---- epilogue: no ring LDS, no barrier; the next tile's first three stages are already in flight ----
    if (!YV4_ABLATE(p.ablate, 8)) {
      unsigned resw[TM * TN][8];
      if (p.res) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            residual_prefetch_h<BF16>(p, lane, m0 + wm * 64 + i * 32, n0 + wn * 64 + jn * 32, resw[jn * TM + i]);
      }
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          epilogue_pairs_h<BF16>(p, acc[i][jn], lane, m0 + wm * 64 + i * 32, n0 + wn * 64 + jn * 32, has2, aff, resw[jn * TM + i]);
    } else if (acc[0][0][0] == 12345.678f) {
      reinterpret_cast<float*>(p.y)[0] = acc[0][1][1] + acc[1][0][2] + acc[1][1][3];      // keep the accumulators live
    }
  }
  if (wm < 2) __builtin_amdgcn_s_barrier();            // same number of barriers for both halves
#ifdef YV4_MEASURE
  if (YV4_ABLATE(p.ablate, 64) && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 101))
    printf("pp3 wg %d wave %d: total %llu | read-issue %llu wait %llu bar1 %llu mfma+dma %llu bar2 %llu between %llu (cycles)\n",
           (int)blockIdx.x, wave, __builtin_amdgcn_s_memtime() - tbegin, tsum[0], tsum[1], tsum[2], tsum[3], tsum[4], tsum[5]);
#endif
#undef YV4_P3_STAGE
#undef YV4_P3_MFMA_J
#undef YV4_P3_LOAD
#undef YV4_P3_ADVANCE
#undef YV4_P3_ISSUE_A
#undef YV4_P3_ISSUE_B
  YV4_P3_WAIT(0);                                      // the zero-filling tail DMAs must land before the LDS is released
#undef YV4_P3_WAIT
}

// Is this layer in the kernel's domain?  3x3, stride 1, pad 1 (so Ho = H, Wo = W), 64-channel chunks of input, even
// Cout (channel pairs are stored as dwords) and dword-aligned output / residual views, 16-bit output.
bool conv3x3_pp_h16_applies(const ConvArgsH& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && (a.Cin & 63) == 0 &&
         !a.ys_on && !a.out_f32 && a.Cout >= 64 && a.Cout <= kP3MaxCout && (a.Cout & 1) == 0 && ((a.y_cs | a.y_co) & 1) == 0 &&
         (a.res == nullptr || ((a.r_cs | a.r_co) & 1) == 0);
}

static int g_p3_cus = 0;

template <bool BF16>
static int launch_p3(const ConvArgsH& a, hipStream_t stream) {
  ConvArgsH p = a;
  const int tiles_m = (p.M + kP3BM - 1) / kP3BM;
  p.tiles_n = (p.Cout + kP3BN - 1) / kP3BN;
  p.fd_hw = make_fastdiv((unsigned)(p.H * p.W));
  p.fd_wo = make_fastdiv((unsigned)p.W);
  const long long tiles = (long long)tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv3x3 pp h16: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  if (g_p3_cus == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
    g_p3_cus = cus;
  }
  const long long xb = (long long)p.N * p.H * p.W * p.x_cs * 2, wb = (long long)p.Cout * p.Kw * 2;
  auto kern = conv3x3_pp_h16_kernel<BF16>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), (size_t)kP3Lds, "conv3x3_pp_h16")) return rc;
  const unsigned grid = (unsigned)(tiles < g_p3_cus ? tiles : g_p3_cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kP3Threads), (size_t)kP3Lds, stream, p, (unsigned)xb, (unsigned)wb, (int)tiles);
  YV4_CHECK_LAUNCH("conv3x3_pp_h16");
  return YV4_OK;
}

int conv3x3_pp_h16_launch(const ConvArgsH& a, bool bf16, hipStream_t s) {
  return bf16 ? launch_p3<true>(a, s) : launch_p3<false>(a, s);
}

}  // namespace yv4
