// Fused implicit-GEMM convolution for gfx950 with 16-bit operands (fp16 or bf16 activations and
// weights, fp32 accumulate on v_mfma_f32_32x32x16_{f16,bf16}, fp32 epilogue, 16-bit or fp32 store).
//
// The reduced-precision form of conv_mfma_f32.hip, for BASELINE.json configs[2..4] (bf16 training,
// fp16 inference): same GEMM view (M = N*Ho*Wo pixels, Ncol = Cout, K = KH*KW*Cin ordered
// (kh,kw,ci)), same fused epilogue (affine1 -> act1 -> +residual -> affine2 -> act2 -> store at a
// channel offset), same LDS image -- a K slice is 64 elements = 128-byte rows, 16-byte chunks XOR-
// swizzled with (row>>1)&7, filled by LDS-DMA (buffer_load_dwordx4 ... lds; out-of-range offsets
// deliver zeros = conv padding, M tail, Cout tail, K tail).  One ds_read_b128 is exactly one
// lane's 8-element operand of a 32x32x16 MFMA (lane (r,h): row r, k = 8h..8h+7 of the step), so a
// slice is 4 MFMA steps per accumulator tile.  The MFMA runs 16x the fp32 rate, so tiles are larger
// (a wave owns up to 64x64 outputs) to keep LDS reads (mt+nt per mt*nt MFMAs) under the array's
// bandwidth.
//
// What the reference does at this precision (mmcv wrap_fp16_model / autocast around ConvModule,
// darknetcsp.py:15-35): conv in half with fp32 accumulation -> round -> BN in fp32 -> round -> Mish ->
// round.  Here the chain after the accumulator stays in fp32 and is rounded once.
#include "conv_h16_common.h"

namespace yv4 {

// GENERAL_K = false: Cin % 64 == 0, a slice lies inside one (kh,kw) tap and the (tap, channel)
//   walk is scalar.  GENERAL_K = true: Cin % 8 == 0 only (stem with C padded to 8, Cin = 32
//   layers, tiny models): every lane derives (tap, channel) of its own 8-element chunk per slice;
//   chunks beyond K read zeros on both operands.
template <bool BF16, int BM, int BN, int WAVES_M, int WAVES_N, bool GENERAL_K, int NBUF>
__global__ __launch_bounds__(kHThreads, 2) void conv_mfma_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes) {
  typedef typename Elem<BF16>::V8 V8;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(NBUF >= 2 && NBUF <= 4, "ring of 2..4 slices");
  constexpr int D = NBUF - 1;   // prefetch distance in K slices
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int PA = BM / 32;
  constexpr int PB = BN / 32;
  constexpr int kDmaPerSlice = PA + PB;
  constexpr int kRowB = 128;    // bytes per LDS row
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_h[];
  char* As = smem_h;                         // [NBUF][BM][128 B]
  char* Bs = smem_h + NBUF * BM * kRowB;     // [NBUF][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int r = lane & 31;
  const int h = lane >> 5;

  const unsigned nwg = gridDim.x;
  const unsigned bid = blockIdx.x;
  const unsigned xcd = bid & 7u, q8 = nwg >> 3, rem8 = nwg & 7u;
  const unsigned tile_s = (xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8) + (bid >> 3);
  const int ksplit = (!GENERAL_K && p.ksplit > 1) ? p.ksplit : 1;
  const unsigned tile = tile_s / (unsigned)ksplit;          // the splits of one tile are neighbours (same L2)
  const int split = (int)(tile_s - tile * (unsigned)ksplit);
  const int tile_n = tile % p.tiles_n;
  const int tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  if (ksplit > 1) p.y = p.ws + (size_t)split * p.M * p.ws_cs;

  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc_h(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_h;

  // staging: this lane fills LDS row (32*q + 8*wave + lane/8), physical chunk lane%8; the logical
  // chunk it must fetch is the same for every pass q ((row>>1)&7 does not change with +32)
  const int srow = 8 * wave + (lane >> 3);
  const int pc = lane & 7;
  const int lc = pc ^ ((srow >> 1) & 7);
  unsigned a_off[PA];
  unsigned long long a_mask[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int m = m0 + srow + 32 * q;
    unsigned long long mk = 0ull;
    unsigned off = 0u;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = fd_div(m, p.fd_hw);
      const int rm = m - n * hw;
      const int ho = fd_div(rm, p.fd_wo);
      const int wo = rm - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad;
      const int wi0 = wo * p.stride - p.pad;
      off = (unsigned)((((int64_t)(n * p.H + hi0) * p.W + wi0) * p.x_cs + p.x_co + (GENERAL_K ? 0 : lc * 8)) * 2);
      mk = tap_mask(hi0, wi0, p.KH, p.KW, p.H, p.W);
    }
    a_off[q] = off;
    a_mask[q] = mk;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int co = n0 + srow + 32 * q;
    b_off[q] = co < p.Cout ? (unsigned)(((int64_t)co * p.Kw + (GENERAL_K ? 0 : lc * 8)) * 2) : kOOB;
  }

  // uniform-K path: channel-chunk-major walk of K (all taps of one 64-channel chunk, then the next chunk), so a tap
  // finds the lines its neighbour taps fetched one or KW slices ago still in L2 (see conv_mfma_f32.hip)
  const int ntaps = p.KH * p.KW;
  int s_tap = 0, s_c0 = 0, s_kh = 0, s_kw = 0;
  unsigned s_kb = 0;
  const int slice0 = ksplit > 1 ? split * p.ks_slices : 0;      // first K slice of this workgroup
  if (slice0) {
    const int chunk = fd_div(slice0, p.fd_taps);
    s_tap = slice0 - chunk * ntaps;
    s_c0 = chunk * kHBK;
    s_kh = fd_div(s_tap, p.fd_kw);
    s_kw = s_tap - s_kh * p.KW;
    s_kb = (unsigned)((s_tap * p.Cin + s_c0) * 2);
  }
  int g_k = lc * 8;     // GENERAL_K: first K index of this lane's chunk in the next slice

#define YV4_H_DMA(BUF)                                                                           \
  {                                                                                              \
    const unsigned la_ = lds_base + (unsigned)(((BUF) * BM + 8 * wave) * kRowB);                  \
    const unsigned lb_ = lds_base + (unsigned)((NBUF * BM + (BUF) * BN + 8 * wave) * kRowB);      \
    if (GENERAL_K) {                                                                             \
      const bool kin = g_k < p.K;                                                                \
      const int tap = kin ? fd_div(g_k, p.fd_cin) : 0;                                           \
      const int c = g_k - tap * p.Cin;                                                           \
      const int kh = fd_div(tap, p.fd_kw);                                                       \
      const int kw = tap - kh * p.KW;                                                            \
      const unsigned step = (unsigned)((((int64_t)kh * p.W + kw) * p.x_cs + c) * 2);             \
      _Pragma("unroll") for (int q = 0; q < PA; ++q) {                                           \
        const bool ok = kin && ((a_mask[q] >> tap) & 1ull);                                      \
        lds_dma16_h(rsA, la_ + 32 * q * kRowB, ok ? a_off[q] + step : kOOB, 0u);                  \
      }                                                                                          \
      _Pragma("unroll") for (int q = 0; q < PB; ++q)                                             \
        lds_dma16_h(rsB, lb_ + 32 * q * kRowB, (kin && b_off[q] != kOOB) ? b_off[q] + (unsigned)(g_k * 2) : kOOB, 0u); \
      g_k += kHBK;                                                                               \
    } else {                                                                                     \
      const unsigned step = (unsigned)((((int64_t)s_kh * p.W + s_kw) * p.x_cs + s_c0) * 2);      \
      _Pragma("unroll") for (int q = 0; q < PA; ++q) {                                           \
        const bool ok = (a_mask[q] >> s_tap) & 1ull;                                             \
        lds_dma16_h(rsA, la_ + 32 * q * kRowB, ok ? a_off[q] + step : kOOB, 0u);                  \
      }                                                                                          \
      _Pragma("unroll") for (int q = 0; q < PB; ++q)                                             \
        lds_dma16_h(rsB, lb_ + 32 * q * kRowB, b_off[q], s_kb);                                   \
      s_tap += 1;                                                                                \
      s_kw += 1;                                                                                 \
      const int wrap_w = s_kw == p.KW ? 1 : 0;                                                   \
      s_kw = wrap_w ? 0 : s_kw;                                                                  \
      s_kh += wrap_w;                                                                            \
      const int wrap_t = s_tap == ntaps ? 1 : 0;                                                 \
      s_tap = wrap_t ? 0 : s_tap;                                                                \
      s_kh = wrap_t ? 0 : s_kh;                                                                  \
      s_c0 += wrap_t ? kHBK : 0;                                                                 \
      s_kb = (unsigned)((s_tap * p.Cin + s_c0) * 2);                                             \
    }                                                                                            \
  }

  // fragment read addresses: row*128 B + ((chunk ^ swz) << 4); MFMA step j reads chunk 2j + h
  unsigned a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * TM * 32 + i * 32 + r;
    a_rd[i] = (unsigned)(row * kRowB + ((((row >> 1) & 7) ^ h) << 4));
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * TN * 32 + i * 32 + r;
    b_rd[i] = (unsigned)(row * kRowB + ((((row >> 1) & 7) ^ h) << 4));
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

// All of a slice's fragments are requested FIRST, then its MFMAs run (fenced: left to itself hipcc puts every
// ds_read_b128 and an s_waitcnt right in front of the MFMA that uses it -- four exposed LDS latencies per slice of 4 x
// 64 matrix-pipe cycles).  YV4_H16_STAGED=0 at build time restores the interleaved form for A/B measurement.
#ifndef YV4_H16_STAGED
#define YV4_H16_STAGED 1
#endif
#define YV4_H_COMPUTE(BUF)                                                                       \
  {                                                                                              \
    const char* as_ = As + (BUF) * BM * kRowB;                                                   \
    const char* bs_ = Bs + (BUF) * BN * kRowB;                                                   \
    if (YV4_H16_STAGED && TM * TN <= 2) {                                                        \
      V8 fa[4][TM], fb[4][TN];                                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
            fa[j][i] = *reinterpret_cast<const V8*>(as_ + (a_rd[i] ^ (unsigned)(j << 5)));       \
        _Pragma("unroll") for (int i = 0; i < TN; ++i)                                           \
            fb[j][i] = *reinterpret_cast<const V8*>(bs_ + (b_rd[i] ^ (unsigned)(j << 5)));       \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
          _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                      \
              acc[i][jn] = Elem<BF16>::mfma(fa[j][i], fb[j][jn], acc[i][jn]);                    \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    } else {                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
      V8 fa[TM], fb[TN];                                                                         \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
          fa[i] = *reinterpret_cast<const V8*>(as_ + (a_rd[i] ^ (unsigned)(j << 5)));            \
      _Pragma("unroll") for (int i = 0; i < TN; ++i)                                             \
          fb[i] = *reinterpret_cast<const V8*>(bs_ + (b_rd[i] ^ (unsigned)(j << 5)));            \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
        _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                        \
            acc[i][jn] = Elem<BF16>::mfma(fa[i], fb[jn], acc[i][jn]);                            \
    }                                                                                            \
    }                                                                                            \
  }

  // wait until at most NEWER later slices of this wave's DMAs are still in flight
#define YV4_H_WAIT(NEWER)                                                                        \
  {                                                                                              \
    if ((NEWER) >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * kDmaPerSlice) : "memory"); \
    else if ((NEWER) == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * kDmaPerSlice) : "memory"); \
    else if ((NEWER) == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 * kDmaPerSlice) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                             \
  }

  const int nk_all = (p.K + kHBK - 1) / kHBK;
  const int nk = ksplit > 1 ? (nk_all - slice0 < p.ks_slices ? nk_all - slice0 : p.ks_slices) : nk_all;
  int issued = 0;        // slices whose DMA has been issued
  int wbuf = 0;          // ring slot the next DMA goes to
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < nk) {
      YV4_H_DMA(wbuf);
      ++issued;
      wbuf = wbuf + 1 == NBUF ? 0 : wbuf + 1;
    }
  }
  YV4_H_WAIT(issued - 1);
  __builtin_amdgcn_s_barrier();
  int rbuf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (issued < nk) {     // slot wbuf was last read before the previous barrier
      if (!YV4_ABLATE(p.ablate, 1)) YV4_H_DMA(wbuf);
      ++issued;
      wbuf = wbuf + 1 == NBUF ? 0 : wbuf + 1;
    }
    __builtin_amdgcn_s_setprio(1);
    if (!YV4_ABLATE(p.ablate, 2)) YV4_H_COMPUTE(rbuf);
    __builtin_amdgcn_s_setprio(0);
    rbuf = rbuf + 1 == NBUF ? 0 : rbuf + 1;
    if (kt + 1 < nk) {
      YV4_H_WAIT(issued - (kt + 2));
      if (!YV4_ABLATE(p.ablate, 4)) __builtin_amdgcn_s_barrier();
    }
  }
#undef YV4_H_WAIT
#undef YV4_H_DMA
#undef YV4_H_COMPUTE

  // BatchNorm statistics of the tile while it is still in registers (identity-epilogue convs only: the stored
  // value is the accumulator rounded to the output type).  A lane owns column r of 16 rows of each 32x32 tile;
  // the two half-waves hold the two row halves.  One double atomic per column per wave, spread over
  // YV4_STATS_REPLICAS copies (indexed by the row tile) so that same-address atomics do not serialise.
  if (p.stats) {
    typedef typename Elem<BF16>::T TS;
    const StatRep rep = stat_rep(p.stats, (unsigned)(tile_m), p.Cout);
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float su = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          const float v = p.out_f32 ? acc[i][jn][e] : (float)(TS)acc[i][jn][e];
          if (m < p.M) { su += v; sq += v * v; }
        }
      }
      su += __shfl_xor(su, 32);
      sq += __shfl_xor(sq, 32);
      const int col = n0 + wn * TN * 32 + jn * 32 + r;
      if (h == 0 && col < p.Cout) {
        stat_add(rep, col, su);
        stat_add(rep, p.Cout + col, sq);
      }
    }
  }
  const bool has2 = p.s2 != nullptr;
  const int ymask = p.out_f32 ? 3 : 7;
  const bool vec_ok = ((p.y_cs | p.y_co) & ymask) == 0 && (p.res == nullptr || ((p.r_cs | p.r_co) & 7) == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* ep = reinterpret_cast<float*>(smem_h) + wave * (32 * 36);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int cb = n0 + wn * TN * 32 + jn * 32;
    const int co = cb + (lane & 3) * 8;
    const bool full = vec_ok && co + 7 < p.Cout;
    AffH af;
    if (full) load_affine_h(p, co, has2, af);        // once per column group, shared by the TM row tiles
#pragma unroll
    for (int i = 0; i < TM; ++i)
      epilogue_tile_h<BF16>(p, acc[i][jn], ep, lane, m0 + wm * TM * 32 + i * 32, cb, full, has2, af);
  }
}

template <bool BF16, int BM, int BN, bool GENERAL_K, int NBUF>
static int launch_h16(const ConvArgsH& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)NBUF * (BM + BN) * 128;
  static_assert(lds >= 4 * 32 * 36 * 4, "epilogue patches must fit the K-loop carve");
  ConvArgsH p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  p.fd_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fastdiv((unsigned)p.Wo);
  p.fd_cin = make_fastdiv((unsigned)p.Cin);
  p.fd_kw = make_fastdiv((unsigned)p.KW);
  p.fd_taps = make_fastdiv((unsigned)(p.KH * p.KW));
  const long long tiles = (long long)tiles_m * p.tiles_n * (p.ksplit > 1 ? p.ksplit : 1);
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv h16: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  const long long xb = (long long)p.N * p.H * p.W * p.x_cs * 2, wb = (long long)p.Cout * p.Kw * 2;
  auto kern = conv_mfma_h16_kernel<BF16, BM, BN, 2, 2, GENERAL_K, NBUF>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), lds, "conv_mfma_h16")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(kHThreads), lds, stream, p, (unsigned)xb, (unsigned)wb);
  YV4_CHECK_LAUNCH("conv_mfma_h16");
  return YV4_OK;
}

template <bool BF16, int NBUF>
static int dispatch_h16_n(const ConvArgsH& a, int shape, bool general, hipStream_t s) {
  switch (shape) {
    case YV4_HTILE_128x128:
      return general ? launch_h16<BF16, 128, 128, true, NBUF>(a, s) : launch_h16<BF16, 128, 128, false, NBUF>(a, s);
    case YV4_HTILE_128x64:
      return general ? launch_h16<BF16, 128, 64, true, NBUF>(a, s) : launch_h16<BF16, 128, 64, false, NBUF>(a, s);
    case YV4_HTILE_64x64:
      return general ? launch_h16<BF16, 64, 64, true, NBUF>(a, s) : launch_h16<BF16, 64, 64, false, NBUF>(a, s);
    default:
      break;
  }
  set_error("conv h16: unknown tile id %d", shape);
  return YV4_E_INVALID;
}

// Deeper rings (NBUF 3: two slices in flight) were measured slower on every YOLOv4-L shape: a
// slice costs ~1.3 us of memory latency against 0.2 us of MFMA, so throughput is set by the bytes in
// flight per CU, and a third slot costs exactly the occupancy it buys; a 4-slot ring that issues a
// whole K <= 256 reduction up front was slower still on the 1x1 layers (64x64: 4112 vs 2975 us over the
// network's 1x1 layers): resident workgroups per CU, not prefetch depth, hide the prologue / epilogue.
template <bool BF16>
static int dispatch_h16(const ConvArgsH& a, int tile, bool general, hipStream_t s) {
  // (a 256x128 workgroup tile -- 128x64 per wave, 0.75 fragment reads per MFMA, one workgroup per CU, 2- or
  // 3-slot ring -- measured 3-4x SLOWER on every YOLOv4-L shape: 128 accumulator registers plus the 12 DMA
  // address sets do not fit the 256-VGPR budget of this 4-wave kernel without spilling)
  return dispatch_h16_n<BF16, 2>(a, tile, general, s);
}

// conv3x3_pp_h16.hip
bool conv3x3_pp_h16_applies(const ConvArgsH& a);
int conv3x3_pp_h16_launch(const ConvArgsH& a, bool bf16, hipStream_t s);

// The persistent ping-pong 3x3 kernel takes the stride-1 3x3 layers with >= 128 input channels whose 256 x 128 tiles
// fill the one-workgroup-per-CU rounds: at least 150 tiles and at least 60 % of the last round's CUs busy.  Measured
// per layer (tools/conv_bench.py --dtype bf16 --tiles 4,2, batch 32, one box): 128->128 @76 76 vs 87 us, 256->256 @38
// 74 vs 78, 512->512 @19 67 vs 71, 128->256 @76 143 vs 167, 256->512 @38 117 vs 147, 512->1024 @19 126 vs 145.  A
// batch-2 plan's 24 tiles stay on the generic tiles, which sum K in the same order and apply the same epilogue
// expressions -- same bits.  YV4_PP3=0 (measurement build) switches it off.
static bool prefer_pp3(const ConvArgsH& a) {
  static const int mode = YV4_ENV_INT("YV4_PP3", 1);
  static const int cus = YV4_ENV_INT("YV4_PP3_CUS", 256);
  if (!mode || !conv3x3_pp_h16_applies(a) || a.Cin < 128) return false;
  const long long tiles = ((long long)a.M + 255) / 256 * ((a.Cout + 127) / 128);
  const long long rounds = (tiles + cus - 1) / cus;
  return tiles >= 150 && tiles * 100 >= rounds * cus * 60;
}

// conv3x3_wide_h16.hip
bool conv3x3_wide_h16_applies(const ConvArgsH& a);
int conv3x3_wide_h16_pick(const ConvArgsH& a, double* rounds_eff);
int conv3x3_wide_h16_launch(const ConvArgsH& a, bool bf16, int shape, hipStream_t s);

// The wide-tile 3x3 kernel takes the layers in its domain with >= 128 input channels and at least YV4_W3_MINOUT outputs
// per CU when one of its tile shapes fills the chip's rounds to within YV4_W3_MAXWASTE percent.  Measured per layer
// against the ping-pong kernel (tools/conv_bench.py --dtype bf16 --tiles 4,5, one box, batch 32 / 64):
// 128->128 @76 71 -> 60 / 133 -> 112 us, 256->256 @38 71 -> 52 / 110 -> 97, 128->256 @76 133 -> 116 / 261 -> 211,
// 256->512 @38 109 -> 98 / 213 -> 182, 512->1024 @19 121 -> 93 / 195 -> 179, 512->512 @19 61 -> 64 (23 104 outputs per CU:
// stays on the ping-pong kernel) / 120 -> 92.  YV4_W3=0 (measurement build) switches it off, YV4_W3_SHAPE forces a shape.
static bool prefer_w3(const ConvArgsH& a) {
  static const int mode = YV4_ENV_INT("YV4_W3", 1);
  static const int waste = YV4_ENV_INT("YV4_W3_MAXWASTE", 25);
  static const int min_out = YV4_ENV_INT("YV4_W3_MINOUT", 32768);
  // (64 input channels -- one chunk, nine K tiles per output tile -- only with a full 128-channel tile of outputs:
  // YOLOv4-S 64 -> 128 @52 at batch 256 143 us against 193 on the 128 x 64 tile, 64 -> 64 122 against 108)
  if (!mode || !conv3x3_wide_h16_applies(a) || a.Cin < 64) return false;
  if ((long long)a.M * a.Cout < 256LL * min_out) return false;
  double eff = 0.0;
  if (conv3x3_wide_h16_pick(a, &eff) < 0) return false;
  // 64 input channels (one chunk, nine K tiles per output tile): whatever the rounds -- the alternatives are the 128 x 64
  // tile and the few-channel kernel, 15-25 % behind on these layers at any fill (64 -> 64 on the 384 x 64 shape: @52 batch
  // 256 92 us against 113, @152 batch 32 94 against 116; 64 -> 128 @52 143 against 193; profiles/r06_w3_bn64.txt)
  if (a.Cin < 128) return true;
  if (eff * 100.0 <= 100.0 + waste) return true;
  // A worse fill still wins where the alternative is the ping-pong kernel, whose 256 x 128 tiles quantise the same way
  // (YOLOv5-L at 640: 256->256 @40 is 800 such tiles = 3.1 rounds for either kernel; train step 1 045 -> 1 063, bf16
  // inference 3 698 -> 3 916 images/s with this rule, YOLOv4-L at 608 unchanged).
  static const int vs_pp = YV4_ENV_INT("YV4_W3_VSPP", 1);
  if (!vs_pp || !prefer_pp3(a)) return false;
  const long long tiles = ((long long)a.M + 255) / 256 * ((a.Cout + 127) / 128);
  const long long rounds = (tiles + 255) / 256;
  const double eff_pp = (double)rounds * 256.0 * (256.0 * 128.0) / ((double)a.M * a.Cout);
  return eff <= eff_pp * 1.15;          // (eff carries the pick's 1.04 / 1.12 weight of the smaller wave tiles)
}

// conv_wide_h16.hip
bool conv_wide_h16_applies(const ConvArgsH& a);
int conv_wide_h16_pick(const ConvArgsH& a, double* rounds_eff);
int conv_wide_h16_launch(const ConvArgsH& a, bool bf16, int shape, hipStream_t s);

// The general wide-tile kernel takes what the 3x3 / stride-1 form and the weight-stationary 1x1 kernel leave: stride-2
// 3x3 layers and 1x1 layers with >= YV4_WIDE_MINCIN input channels, when the layer has >= YV4_WIDE_MINOUT outputs per CU
// and a tile shape fills the rounds to within YV4_WIDE_MAXWASTE percent (YV4_WIDE=0 switches it off).
static bool prefer_wide(const ConvArgsH& a) {
  static const int mode = YV4_ENV_INT("YV4_WIDE", 1);
  static const int waste = YV4_ENV_INT("YV4_WIDE_MAXWASTE", 35);     // 25 -> 35: YOLOv5-L 640 (256->256 1x1 @40 = 200 tiles of 256 x 256), DESIGN 11.5f
  static const int min_out = YV4_ENV_INT("YV4_WIDE_MINOUT", 32768);
  static const int min_cin = YV4_ENV_INT("YV4_WIDE_MINCIN", 256);
  if (!mode || !conv_wide_h16_applies(a)) return false;
  const bool s2 = a.KH == 3 && a.KW == 3 && a.stride == 2 && a.Cin >= 64;
  const bool pw = a.KH == 1 && a.KW == 1 && a.Cin >= min_cin;
  if (!(s2 || pw || a.ys_on)) return false;
  if ((long long)a.M * a.Cout < 256LL * min_out || a.K < 256) return false;
  double eff = 0.0;
  if (conv_wide_h16_pick(a, &eff) < 0) return false;
  return eff * 100.0 <= 100.0 + waste;
}

// conv1x1_ws_h16.hip
bool conv1x1_ws_applies(const ConvArgsH& a);
int conv1x1_ws_launch(const ConvArgsH& a, bool bf16, hipStream_t s);

// The weight-stationary pointwise kernel takes the 1x1 layers in its domain with at least YV4_WS_MINSTRIPS strips of 32
// pixels for its 2 048 persistent waves (YV4_WS=0 switches it off).  Measured, 256 -> 256: 1 444 strips (38 x 38 x 32) 20 us
// against the generic tile's 19; 2 888 (38 x 38 x 64, the training batch) 32 against 37; 5 776 (76 x 76 x 32) and up: 1.3-1.5x.
static bool prefer_ws(const ConvArgsH& a) {
  static const int mode = YV4_ENV_INT("YV4_WS", 1);
  static const int min_strips = YV4_ENV_INT("YV4_WS_MINSTRIPS", 2560);
  if (!mode || !conv1x1_ws_applies(a)) return false;
  return ((long long)a.M + 31) / 32 >= (long long)min_strips;
}

// conv3x3_small_h16.hip
bool conv3x3_small_applies(const ConvArgsH& a);
int conv3x3_small_launch(const ConvArgsH& a, bool bf16, hipStream_t s);

// The few-channel 3x3 kernel takes the layers in its domain with at least YV4_S3_MINTILES 16 x 16 output tiles
// (YV4_S3=0 switches it off).
static bool prefer_s3(const ConvArgsH& a) {
  static const int mode = YV4_ENV_INT("YV4_S3", 1);
  static const int min_tiles = YV4_ENV_INT("YV4_S3_MINTILES", 1024);
  if (!mode || !conv3x3_small_applies(a)) return false;
  const long long ty = (a.Ho + 15) / 16, tx = (a.Wo + 15) / 16;
  // at 64 input channels the generic tiles are close (0.9-1.2x, measured): only maps that fill their 16 x 16 tiles
  if (a.Cin == 64 && (long long)a.Ho * a.Wo * 100 < ty * tx * 256 * 85) return false;
  return (long long)a.N * ty * tx >= min_tiles;
}

static int pick_tile_h16(long long M, int Cout, long long K) {
  // From the per-layer table of tools/conv_bench.py --dtype bf16 (YOLOv4-L, batch 32; after the prologue /
  // epilogue work of round 1 the 128x64 tile -- three workgroups per CU -- is the best or within 2 % of the best
  // on every shape but the widest deep reductions with several rounds of tiles).
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
  if (K >= 2304 && Cout >= 512 && tiles(128, 128) >= 1024) return YV4_HTILE_128x128;
  if (tiles(128, 64) >= 256) return YV4_HTILE_128x64;
  return YV4_HTILE_64x64;
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_conv_h16_pick_tile(const yv4_conv_desc* d) {
  if (!d) return YV4_TILE_AUTO;
  {
    ConvArgsH a{};
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.Cin = d->Cin; a.Cout = d->Cout; a.ys_on = 0; a.M = (int)((long long)d->N * d->Ho * d->Wo);
    a.ksplit = 0; a.ws = nullptr;
    a.K = a.Kw = d->KH * d->KW * d->Cin; a.res = nullptr; a.out_f32 = (d->Cout & 1) ? 1 : 0;   // (odd Cout: a pred map)
    a.y_cs = d->y_cstride; a.y_co = d->y_coff; a.r_cs = d->r_cstride; a.r_co = d->r_coff;
    if (prefer_w3(a)) return YV4_HTILE_W3x3;
    if (prefer_pp3(a)) return YV4_HTILE_PP3x3;
    a.K = a.Kw;
    if (prefer_wide(a)) return YV4_HTILE_WIDE;
    if (prefer_ws(a)) return YV4_HTILE_WS_1x1;
    a.N = d->N; a.stats = nullptr; a.y_cs = d->y_cstride; a.y_co = d->y_coff; a.r_cs = d->r_cstride; a.r_co = d->r_coff;
    if (prefer_s3(a)) return YV4_HTILE_S3x3;
  }
  return pick_tile_h16((long long)d->N * d->Ho * d->Wo, d->Cout, (long long)d->KH * d->KW * d->Cin);
}

static int conv_h16_impl(const yv4_conv_desc* d, int dtype, int out_dtype, const void* x, const void* w,
                         const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                         const void* residual, void* y, double* stats, void* stream) {
  YV4_REQUIRE(d && x && w && scale1 && shift1 && y, "conv h16: null argument");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "conv h16: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(out_dtype == dtype || out_dtype == YV4_F32, "conv h16: out_dtype must be the operand type or YV4_F32");
  YV4_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), "conv h16: scale2/shift2 must come together");
  YV4_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv h16: empty shape");
  YV4_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv h16: bad kernel/stride/pad");
  YV4_REQUIRE(d->KH * d->KW <= 64, "conv h16: kernels above 64 taps are not supported");
  YV4_REQUIRE(d->Cin % 8 == 0 && d->x_cstride % 8 == 0 && d->x_coff % 8 == 0,
              "conv h16: Cin (%d), x_cstride (%d), x_coff (%d) must be multiples of 8", d->Cin, d->x_cstride, d->x_coff);
  YV4_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv h16: x / w must be 16-byte aligned");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride, "conv h16: input view exceeds its pixel stride");
  YV4_REQUIRE(d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride, "conv h16: output view exceeds its pixel stride");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  YV4_REQUIRE(Ho == d->Ho && Wo == d->Wo, "conv h16: Ho/Wo (%d,%d) do not match the geometry (%d,%d)", d->Ho, d->Wo, Ho,
              Wo);
  if (residual)
    YV4_REQUIRE(d->r_coff >= 0 && d->r_coff + d->Cout <= d->r_cstride, "conv h16: residual view exceeds its pixel stride");
  YV4_REQUIRE(d->act1 >= 0 && d->act1 <= YV4_ACT_SWISH && d->act2 >= 0 && d->act2 <= YV4_ACT_SWISH,
              "conv h16: unknown activation id");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  YV4_REQUIRE(M < (1LL << 31), "conv h16: N*Ho*Wo = %lld does not fit 31 bits", M);
  YV4_REQUIRE((long long)d->N * d->H * d->W < (1LL << 31), "conv h16: N*H*W does not fit 31 bits");
  const long long K = (long long)d->KH * d->KW * d->Cin;
  YV4_REQUIRE((long long)d->N * d->H * d->W * d->x_cstride * 2 < 0xFFFFFFF0LL && (long long)d->Cout * K * 2 < 0xFFFFFFF0LL,
              "conv h16: tensors of 4 GiB or more are not addressable through a buffer descriptor");

  ConvArgsH a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = scale2; a.t2 = shift2;
  a.res = residual; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff;
  a.r_cs = d->r_cstride; a.r_co = d->r_coff;
  a.act1 = d->act1; a.act2 = d->act2; a.slope1 = d->slope1; a.slope2 = d->slope2;
  a.M = (int)M; a.K = (int)K; a.Kw = (int)K; a.tiles_n = 0;
  a.out_f32 = out_dtype == YV4_F32 ? 1 : 0;
  a.ys_on = 0;
  a.ksplit = 0; a.ks_slices = 0; a.ws_cs = 0; a.ws = nullptr;
  static const int ablate = YV4_ENV_INT("YV4_H16_ABLATE", 0);
  a.ablate = ablate;
  a.nt_out = (d->flags & YV4_CONV_NT_OUT) ? 1 : 0;
  a.stats = stats;
  const bool general = (d->Cin % kHBK) != 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int w3_forced = (d->tile > 8 && (d->tile & 7) == YV4_HTILE_W3x3 && d->tile <= YV4_HTILE_W3x3_SHAPE(6)) ? (d->tile >> 3) - 1 : -1;
  if (d->tile == YV4_HTILE_W3x3 || w3_forced >= 0)
    YV4_REQUIRE(conv3x3_wide_h16_applies(a), "conv h16: the wide 3x3 tile needs a 3x3 / stride 1 / pad 1 conv with Cin %% 64 == 0, "
                "Cout %% 16 == 0 (64 .. 1024), 16-bit output and 8-aligned channel strides / offsets");
  if (d->tile == YV4_HTILE_W3x3 || w3_forced >= 0 || (d->tile == YV4_TILE_AUTO && prefer_w3(a))) {
    static const int shape = YV4_ENV_INT("YV4_W3_SHAPE", -1);
    return conv3x3_wide_h16_launch(a, dtype == YV4_BF16, w3_forced >= 0 ? w3_forced : shape, s);
  }
  const int wide_forced = (d->tile >= YV4_HTILE_WIDE_SHAPE(0) && (d->tile & 7) == 0 && d->tile <= YV4_HTILE_WIDE_SHAPE(4)) ? (d->tile >> 3) - 2 : -1;
  if (d->tile == YV4_HTILE_WIDE || wide_forced >= 0)
    YV4_REQUIRE(conv_wide_h16_applies(a), "conv h16: the wide general tile needs Cin %% 64 == 0, Cout %% 16 == 0 (64 .. 2048), 16-bit "
                "output and 8-aligned channel strides / offsets");
  if (d->tile == YV4_HTILE_WIDE || wide_forced >= 0 || (d->tile == YV4_TILE_AUTO && !prefer_pp3(a) && prefer_wide(a))) {
    static const int shape = YV4_ENV_INT("YV4_WIDE_SHAPE", -1);
    return conv_wide_h16_launch(a, dtype == YV4_BF16, wide_forced >= 0 ? wide_forced : shape, s);
  }
  if (d->tile == YV4_HTILE_PP3x3)
    YV4_REQUIRE(conv3x3_pp_h16_applies(a), "conv h16: the ping-pong 3x3 tile needs a 3x3 / stride 1 / pad 1 conv with "
                "Cin %% 64 == 0, even Cout >= 64, 16-bit output and even channel strides / offsets");
  if (d->tile == YV4_HTILE_PP3x3 || (d->tile == YV4_TILE_AUTO && prefer_pp3(a))) return conv3x3_pp_h16_launch(a, dtype == YV4_BF16, s);
  if (d->tile == YV4_HTILE_WS_1x1)
    YV4_REQUIRE(conv1x1_ws_applies(a), "conv h16: the weight-stationary tile needs a 1x1 / stride 1 conv with Cin <= 256, "
                "Cout >= 16 (even unless the output is fp32); a residual needs a 16-bit output");
  if (d->tile == YV4_HTILE_WS_1x1 || (d->tile == YV4_TILE_AUTO && prefer_ws(a))) return conv1x1_ws_launch(a, dtype == YV4_BF16, s);
  if (d->tile == YV4_HTILE_S3x3)
    YV4_REQUIRE(conv3x3_small_applies(a), "conv h16: the few-channel 3x3 tile needs a 3x3 / stride 1 / pad 1 conv with Cin 16, 32 "
                "or 64, even Cout in [16, 64] and 16-bit output");
  if (d->tile == YV4_HTILE_S3x3 || (d->tile == YV4_TILE_AUTO && prefer_s3(a))) return conv3x3_small_launch(a, dtype == YV4_BF16, s);
  const int tile = d->tile == YV4_TILE_AUTO ? pick_tile_h16(M, d->Cout, K) : d->tile;
  return dtype == YV4_BF16 ? dispatch_h16<true>(a, tile, general, s) : dispatch_h16<false>(a, tile, general, s);
}

extern "C" int yv4_conv_bn_act_fwd_h16(const yv4_conv_desc* d, int dtype, int out_dtype, const void* x, const void* w,
                                       const float* scale1, const float* shift1, const float* scale2,
                                       const float* shift2, const void* residual, void* y, void* stream) {
  return conv_h16_impl(d, dtype, out_dtype, x, w, scale1, shift1, scale2, shift2, residual, y, nullptr, stream);
}

// Pure convolution (identity epilogue) that also leaves the BatchNorm statistics of its output in `stats`.
int conv_stats_h16(const yv4_conv_desc* d, int dtype, const void* x, const void* w, const float* ones, const float* zeros,
                   void* y, double* stats, void* stream) {
  return conv_h16_impl(d, dtype, dtype, x, w, ones, zeros, nullptr, nullptr, nullptr, y, stats, stream);
}

// ---- split-K: the single-image (latency) form of the 16-bit tiles (the fp32 form: conv_mfma_f32.hip) ----------------
// At batch 1 (the reference's only published protocol, tools/analysis_tools/benchmark.py:83-109) the deep layers have a
// handful of output tiles and 36-72 K slices each.  Splitting K over `ksplit` workgroups per 64 x 64 tile fills the CUs;
// partials go to per-split fp32 slabs (plain stores, no atomics) that one small kernel adds IN SLAB ORDER before the
// usual epilogue (the expressions of epilogue_tile_h), so the result is deterministic -- but its summation order is not
// the unsplit tiles': plans use it for N == 1 only, where no cross-batch bit-exactness is claimed.
namespace yv4 {
template <bool BF16>
__global__ __launch_bounds__(256) void splitk_finish_h16_kernel(ConvArgsH p) {
  typedef typename Elem<BF16>::T T;
  typedef typename Elem<BF16>::V8 V8;
  const int c8n = p.ws_cs >> 3;
  const long long total = (long long)p.M * c8n;
  const bool has2 = p.s2 != nullptr;
  const size_t slab = (size_t)p.M * p.ws_cs;
  const bool vec_y = p.out_f32 ? ((p.y_cs | p.y_co) & 3) == 0 : ((p.y_cs | p.y_co) & 7) == 0;
  const bool vec_r = p.res == nullptr || ((p.r_cs | p.r_co) & 7) == 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / c8n);
    const int co = (int)(i - (long long)m * c8n) * 8;
    const float* src = p.ws + (int64_t)m * p.ws_cs + co;
    float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
    for (int s = 1; s < p.ksplit; ++s) {
      const float4 b0 = *reinterpret_cast<const float4*>(src + s * slab), b1 = *reinterpret_cast<const float4*>(src + s * slab + 4);
      a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
      a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    }
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const bool full = co + 7 < p.Cout;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = full ? co + u : (co + u < p.Cout ? co + u : 0);
      v[u] = __builtin_fmaf(v[u], p.s1[c], p.t1[c]);
    }
    act_row8(v, p.act1, p.slope1);
    if (p.res) {
      const T* rp = reinterpret_cast<const T*>(p.res) + (int64_t)m * p.r_cs + p.r_co + co;
      if (full && vec_r) {
        const V8 rr = *reinterpret_cast<const V8*>(rp);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] += (float)rr[u];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (co + u < p.Cout) v[u] += (float)rp[u];
      }
    }
    if (has2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = full ? co + u : (co + u < p.Cout ? co + u : 0);
        v[u] = __builtin_fmaf(v[u], p.s2[c], p.t2[c]);
      }
      act_row8(v, p.act2, p.slope2);
    }
    if (p.out_f32) {
      float* dst = reinterpret_cast<float*>(p.y) + (int64_t)m * p.y_cs + p.y_co + co;
      if (full && vec_y) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (co + u < p.Cout) dst[u] = v[u];
      }
    } else {
      T* dst = reinterpret_cast<T*>(p.y) + (int64_t)m * p.y_cs + p.y_co + co;
      if (full && vec_y) {
        V8 o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = (T)v[u];
        *reinterpret_cast<V8*>(dst) = o;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (co + u < p.Cout) dst[u] = (T)v[u];
      }
    }
  }
}

// how many ways to split K on a 256-CU chip: double while the 64 x 64 tiles x splits stay under ~2 per CU and every split
// keeps at least 4 slices of 64; 1 = do not split (enough tiles already, or a layer outside the uniform-K tiles)
static int splitk_choice_h16(const yv4_conv_desc* d) {
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const long long K = (long long)d->KH * d->KW * d->Cin;
  if (d->Cin % kHBK != 0 || M >= (1LL << 31)) return 1;
  if ((long long)d->N * d->H * d->W * d->x_cstride * 2 >= 0xFFFFFFF0LL || (long long)d->Cout * K * 2 >= 0xFFFFFFF0LL) return 1;
  const long long tiles = ((M + 63) / 64) * ((d->Cout + 63) / 64);
  const int nk = (int)(K / kHBK);
  static const int target = YV4_ENV_INT("YV4_SPLITK_TARGET_H16", 512);
  static const int min_slices = YV4_ENV_INT("YV4_SPLITK_MINSL_H16", 4);
  int ks = 1;
  while (tiles * ks < target && nk / (ks * 2) >= min_slices && ks < 32) ks *= 2;
  return ks;
}
}  // namespace yv4

extern "C" size_t yv4_conv_h16_splitk_workspace(const yv4_conv_desc* d, int* ksplit) {
  if (ksplit) *ksplit = 1;
  if (!d) return 0;
  const int ks = splitk_choice_h16(d);
  if (ksplit) *ksplit = ks;
  if (ks <= 1) return 0;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  return (size_t)ks * (size_t)M * (size_t)((d->Cout + 7) / 8 * 8) * sizeof(float);
}

extern "C" int yv4_conv_bn_act_fwd_h16_splitk(const yv4_conv_desc* d, int dtype, int out_dtype, const void* x, const void* w,
                                              const float* scale1, const float* shift1, const float* scale2,
                                              const float* shift2, const void* residual, void* y, float* workspace,
                                              size_t workspace_bytes, void* stream) {
  YV4_REQUIRE(d, "conv h16 splitk: null descriptor");
  const int ks = splitk_choice_h16(d);
  if (ks <= 1) return yv4_conv_bn_act_fwd_h16(d, dtype, out_dtype, x, w, scale1, shift1, scale2, shift2, residual, y, stream);
  YV4_REQUIRE(x && w && scale1 && shift1 && y && workspace, "conv h16 splitk: null argument");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "conv h16 splitk: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(out_dtype == dtype || out_dtype == YV4_F32, "conv h16 splitk: out_dtype must be the operand type or YV4_F32");
  YV4_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), "conv h16 splitk: scale2/shift2 must come together");
  YV4_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 64 &&
              d->stride > 0 && d->pad >= 0, "conv h16 splitk: bad shape");
  YV4_REQUIRE(d->x_cstride % 8 == 0 && d->x_coff % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 &&
              ((uintptr_t)workspace & 15) == 0, "conv h16 splitk: alignment");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride,
              "conv h16 splitk: view exceeds its pixel stride");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  YV4_REQUIRE(Ho == d->Ho && Wo == d->Wo, "conv h16 splitk: Ho/Wo do not match the geometry");
  if (residual) YV4_REQUIRE(d->r_coff >= 0 && d->r_coff + d->Cout <= d->r_cstride, "conv h16 splitk: residual view");
  YV4_REQUIRE(d->act1 >= 0 && d->act1 <= YV4_ACT_SWISH && d->act2 >= 0 && d->act2 <= YV4_ACT_SWISH, "conv h16 splitk: activation id");
  YV4_REQUIRE(workspace_bytes >= yv4_conv_h16_splitk_workspace(d, nullptr), "conv h16 splitk: workspace too small");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  ConvArgsH a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = scale2; a.t2 = shift2; a.res = residual; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff;
  a.r_cs = d->r_cstride; a.r_co = d->r_coff;
  a.act1 = d->act1; a.act2 = d->act2; a.slope1 = d->slope1; a.slope2 = d->slope2;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin; a.Kw = a.K; a.tiles_n = 0; a.ys_on = 0; a.stats = nullptr; a.ablate = 0; a.nt_out = 0;
  a.out_f32 = out_dtype == YV4_F32 ? 1 : 0;
  const int nk = a.K / kHBK;
  a.ks_slices = (nk + ks - 1) / ks;
  a.ksplit = (nk + a.ks_slices - 1) / a.ks_slices;      // no empty split (72 slices 16 ways = 15 splits of 5, the last of 2)
  a.ws_cs = (d->Cout + 7) / 8 * 8; a.ws = workspace;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (int rc = dtype == YV4_BF16 ? launch_h16<true, 64, 64, false, 2>(a, s) : launch_h16<false, 64, 64, false, 2>(a, s)) return rc;
  const long long work = M * (a.ws_cs / 8);
  unsigned g = (unsigned)((work + 255) / 256);
  if (g > 2048) g = 2048;
  if (dtype == YV4_BF16) hipLaunchKernelGGL(splitk_finish_h16_kernel<true>, dim3(g), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(splitk_finish_h16_kernel<false>, dim3(g), dim3(256), 0, s, a);
  YV4_CHECK_LAUNCH("conv h16 splitk finish");
  return YV4_OK;
}

// 16-bit form of yv4_conv_scatter_fwd (conv_mfma_f32.hip): one parity class of a stride-2 data gradient.
extern "C" int yv4_conv_scatter_fwd_h16(const yv4_conv_desc* d, int dtype, const void* x, const void* w,
                                        const float* scale1, const float* shift1, void* y, int Hy, int Wy, int sh, int sw,
                                        int oh, int ow, void* stream) {
  YV4_REQUIRE(d && x && w && scale1 && shift1 && y, "conv scatter h16: null argument");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "conv scatter h16: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0,
              "conv scatter h16: empty shape");
  YV4_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 64 && d->stride == 1 && d->pad >= 0,
              "conv scatter h16: stride-1 kernels only");
  YV4_REQUIRE(d->Cin % 8 == 0 && d->x_cstride % 8 == 0 && d->x_coff % 8 == 0 && d->y_cstride % 8 == 0 && d->y_coff % 8 == 0,
              "conv scatter h16: channel counts / strides / offsets must be multiples of 8");
  YV4_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv scatter h16: x / w must be 16-byte aligned");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride,
              "conv scatter h16: view exceeds its pixel stride");
  YV4_REQUIRE(sh > 0 && sw > 0 && oh >= 0 && ow >= 0 && (d->Ho - 1) * sh + oh < Hy && (d->Wo - 1) * sw + ow < Wy,
              "conv scatter h16: the scattered grid does not fit the output tensor");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const long long K = (long long)d->KH * d->KW * d->Cin;
  YV4_REQUIRE(M < (1LL << 31) && (long long)d->N * d->H * d->W * d->x_cstride * 2 < 0xFFFFFFF0LL &&
              (long long)d->Cout * K * 2 < 0xFFFFFFF0LL, "conv scatter h16: tensors of 4 GiB or more are not supported");
  ConvArgsH a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = nullptr; a.t2 = nullptr; a.res = nullptr; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = 1; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff; a.r_cs = 0; a.r_co = 0;
  a.act1 = 0; a.act2 = 0; a.slope1 = 0.f; a.slope2 = 0.f; a.stats = nullptr;
  a.M = (int)M; a.K = (int)K; a.Kw = (int)K; a.tiles_n = 0; a.out_f32 = 0; a.ablate = 0; a.nt_out = (d->flags & YV4_CONV_NT_OUT) ? 1 : 0;
  a.ksplit = 0; a.ks_slices = 0; a.ws_cs = 0; a.ws = nullptr;
  a.ys_on = 1; a.ys_H = Hy; a.ys_W = Wy; a.ys_sh = sh; a.ys_sw = sw; a.ys_oh = oh; a.ys_ow = ow;
  const bool general = (d->Cin % kHBK) != 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // the classes of the wide stride-2 layers' data gradients: the general wide-tile kernel (same bits as the generic tiles)
  if (d->tile == YV4_HTILE_WIDE || (d->tile == YV4_TILE_AUTO && prefer_wide(a))) {
    YV4_REQUIRE(conv_wide_h16_applies(a), "conv scatter h16: the wide tile needs Cin %% 64 == 0, Cout %% 16 == 0 and 8-aligned views");
    return conv_wide_h16_launch(a, dtype == YV4_BF16, -1, s);
  }
  const int tile = d->tile == YV4_TILE_AUTO ? pick_tile_h16(M, d->Cout, K) : d->tile;
  return dtype == YV4_BF16 ? dispatch_h16<true>(a, tile, general, s) : dispatch_h16<false>(a, tile, general, s);
}
