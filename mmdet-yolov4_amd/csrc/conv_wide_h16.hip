// General fused convolution on WIDE wave tiles (v_mfma_f32_16x16x32, 16 PT pixels x 64 channels per wave) for gfx950 with
// 16-bit operands: tile id YV4_HTILE_WIDE (round 4).  The wave tile, operand roles (weights = the MFMA's A operand with
// the channel permutation that gives a lane 16 consecutive channels of one pixel), LDS swizzles, fragment schedule and
// epilogue are conv3x3_wide_h16.hip's; what differs is the pixel operand: a K tile here is ONE (64-channel chunk, tap)
// whose BM x 64 pixel tile is gathered by LDS-DMA with the tap's offset -- conv padding, image borders and the M tail
// are out-of-range buffer offsets (zeros), so fragment reads need no masks -- which makes it the implicit GEMM of ANY
// kernel size / stride with Cin % 64 == 0: the stride-2 3x3 layers of CSPDarknet53 / PAN
// (mmdet/models/backbones/darknetcsp.py:262-335), the 1x1 layers with deep reductions or wide outputs, and the scattered
// parity classes of the stride-2 data gradients (output rows through ConvArgsH's row map).  A and B tiles of the next K
// tile are issued at the start of a K tile into the other LDS slot, confirmed by this wave's vmcnt(0) at its end,
// published by ONE workgroup barrier per K tile; the issue side runs on across tiles of the persistent grid.
// K order (chunk-major, taps inside) and epilogue expressions are the generic tiles': the same bits
// (tests/test_gpu_h16.py::test_wide_general_matches_generic_bitwise).
#include "conv_h16_common.h"
#include "conv_wide_common.h"

namespace yv4 {

template <bool BF16> struct MfmaW;
template <> struct MfmaW<true> {
  static __device__ __forceinline__ wide_acc_t run(bf16x8 a, bf16x8 b, wide_acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaW<false> {
  static __device__ __forceinline__ wide_acc_t run(f16x8 a, f16x8 b, wide_acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};


template <bool BF16, int PT, int WAVES_M>
__global__ __launch_bounds__(kWideThreads, 2) void conv_wide_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef typename Elem<BF16>::V8 V8;
  typedef WideGeom<PT, WAVES_M, false> G_;
  constexpr int WAVES_N = G_::WAVES_N, BN = G_::BN, BM = G_::BM, WMr = G_::WMr, QA = G_::QA, PB = G_::PB;
  constexpr int PH = PT / 2;
  constexpr int kRowB = 128;
  static_assert(BM % 64 == 0, "whole DMA passes");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_wg[];
  char* As = smem_wg;                        // [2][BM][128 B]
  char* Bs = smem_wg + 2 * G_::ABytes;       // [2][BN][128 B]
  float* aff = reinterpret_cast<float*>(smem_wg + G_::RingBytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int fr = lane & 15;
  const int fq = lane >> 4;

  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc_h(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_wg;
  const int nwg = (int)gridDim.x;

  const unsigned q8 = (unsigned)ntiles >> 3, rem8 = (unsigned)ntiles & 7u;
  auto tile_of = [&](int vt) -> unsigned {
    const unsigned x = (unsigned)vt & 7u;
    return (x < rem8 ? x * (q8 + 1) : rem8 * (q8 + 1) + (x - rem8) * q8) + ((unsigned)vt >> 3);
  };

  // ---- staging lanes (the tile of the NEXT K tile) ----
  const int srow = 8 * wave + (lane >> 3);
  const int pc = lane & 7;
  const int lcA = pc ^ ((srow >> 1) & 7);
  const int lcB = pc ^ wide_swz_b(srow);
  unsigned a_off[QA];
  unsigned long long a_mask[QA];
  unsigned b_off[PB];
  auto issue_tile_setup = [&](int vt) {
    const bool live = vt < ntiles;
    const unsigned tile = live ? tile_of(vt) : 0u;
    const int tn = (int)(tile % (unsigned)p.tiles_n);
    const int m0i = (int)(tile / (unsigned)p.tiles_n) * BM;
    const int n0i = tn * BN;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int m = m0i + srow + 64 * q;
      unsigned long long mk = 0ull;
      unsigned off = 0u;
      if (live && m < p.M) {
        const int hw = p.Ho * p.Wo;
        const int n = fd_div(m, p.fd_hw);
        const int rm = m - n * hw;
        const int ho = fd_div(rm, p.fd_wo);
        const int wo = rm - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad;
        const int wi0 = wo * p.stride - p.pad;
        off = (unsigned)((((int64_t)(n * p.H + hi0) * p.W + wi0) * p.x_cs + p.x_co + lcA * 8) * 2);
        mk = tap_mask(hi0, wi0, p.KH, p.KW, p.H, p.W);
      }
      a_off[q] = off;
      a_mask[q] = mk;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int co = n0i + srow + 64 * q;
      b_off[q] = (live && co < p.Cout) ? (unsigned)(((int64_t)co * p.Kw + lcB * 8) * 2) : kOOB;
    }
  };

  // ---- fragment read addresses ----
  unsigned a_rd[2], w_rd[2];
  {
    const int row = wm * WMr + fr;
    const int rw = wn * 64 + 16 * (fr >> 2) + (fr & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_rd[ks] = (unsigned)(row * kRowB + (((fq + 4 * ks) ^ ((row >> 1) & 7)) << 4));
      w_rd[ks] = (unsigned)(rw * kRowB + (((fq + 4 * ks) ^ wide_swz_b(rw)) << 4));
    }
  }

  const int nchunks = p.Cin >> 6;
  const int ntaps = p.KH * p.KW;
  const int NK = nchunks * ntaps;            // K tiles per output tile (chunk-major, taps inside)

  const bool has2 = p.s2 != nullptr;

  // issue-side walk: the K tile after the one being computed
  int n_vt = (int)blockIdx.x, n_k = 0, n_tap = 0, n_kh = 0, n_kw = 0, n_c0 = 0;
#define YV4_WG_ISSUE(SLOT)                                                                          \
  {                                                                                                 \
    const unsigned la_ = lds_base + (unsigned)((SLOT) * G_::ABytes + 8 * wave * kRowB);              \
    const unsigned lb_ = lds_base + (unsigned)(2 * G_::ABytes + (SLOT) * G_::BBytes + 8 * wave * kRowB); \
    const unsigned step_ = (unsigned)((((int64_t)n_kh * p.W + n_kw) * p.x_cs + n_c0) * 2);           \
    const unsigned kb_ = (unsigned)((n_tap * p.Cin + n_c0) * 2);                                     \
    _Pragma("unroll") for (int q = 0; q < PB; ++q) lds_dma16_h(rsB, lb_ + 64 * q * kRowB, b_off[q], kb_); \
    _Pragma("unroll") for (int q = 0; q < QA; ++q) {                                                \
      const bool ok_ = (a_mask[q] >> n_tap) & 1ull;                                                 \
      lds_dma16_h(rsA, la_ + 64 * q * kRowB, ok_ ? a_off[q] + step_ : kOOB, 0u);                     \
    }                                                                                               \
    n_k += 1; n_tap += 1; n_kw += 1;                                                                \
    if (n_kw == p.KW) { n_kw = 0; n_kh += 1; }                                                      \
    if (n_tap == ntaps) { n_tap = 0; n_kh = 0; n_c0 += kHBK; }                                      \
    if (n_k == NK) {                                                                                \
      n_k = 0; n_c0 = 0;                                                                            \
      n_vt += nwg;                                                                                  \
      issue_tile_setup(n_vt);                                                                       \
    }                                                                                               \
  }

  issue_tile_setup(n_vt);
  YV4_WG_ISSUE(0);
  // the layer's affine into LDS behind the first fills' issue: its memory round trip runs beside theirs
  for (int c = tid; c < p.Cout; c += kWideThreads) {
    aff[c] = p.s1[c];
    aff[p.Cout + c] = p.t1[c];
    aff[2 * p.Cout + c] = has2 ? p.s2[c] : 1.f;
    aff[3 * p.Cout + c] = has2 ? p.t2[c] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // (also publishes the affine)

  unsigned T_ = 0u;
  for (int vt = (int)blockIdx.x; vt < ntiles; vt += nwg) {
    const unsigned tile = tile_of(vt);
    const int tile_n = (int)(tile % (unsigned)p.tiles_n);
    const int tile_m = (int)(tile / (unsigned)p.tiles_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    wide_acc_t acc[PT][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[pt][t] = wide_acc_t{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < NK; ++kt) {
      const unsigned slot = T_ & 1u;
      const char* as_ = As + slot * G_::ABytes;
      const char* bs_ = Bs + slot * G_::BBytes;
      YV4_WG_ISSUE(slot ^ 1u);
      V8 wf[4][2], pf[PH][2];
      // ---- phase 1
#pragma unroll
      for (int t = 0; t < 4; ++t)      // all four channel tiles now: phase 2 starts without an LDS round trip
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const V8*>(bs_ + w_rd[ks] + t * 512);
#pragma unroll
      for (int i = 0; i < PH; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pf[i][ks] = *reinterpret_cast<const V8*>(as_ + a_rd[ks] + i * 2048);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < PH; ++i) acc[i][t] = MfmaW<BF16>::run(wf[t][ks], pf[i][ks], acc[i][t]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 2 (fragments already in registers)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 2; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < PH; ++i) acc[i][t] = MfmaW<BF16>::run(wf[t][ks], pf[i][ks], acc[i][t]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 3
#pragma unroll
      for (int i = 0; i < PH; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pf[i][ks] = *reinterpret_cast<const V8*>(as_ + a_rd[ks] + (PH + i) * 2048);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 2; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < PH; ++i) acc[PH + i][t] = MfmaW<BF16>::run(wf[t][ks], pf[i][ks], acc[PH + i][t]);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase 4
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < PH; ++i) acc[PH + i][t] = MfmaW<BF16>::run(wf[t][ks], pf[i][ks], acc[PH + i][t]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      T_ += 1u;
    }

    // ---- epilogue (conv_wide_common.h): lane (fr, fq) owns pixel m0 + wm WMr + 16 pt + fr, channels cl .. cl + 15 ----
    wide_epilogue_h16<BF16, PT, true>(p, aff, has2, acc, m0 + wm * WMr + fr, n0 + wn * 64 + 16 * fq, lane,
                                      (unsigned)(tile_m * WAVES_M + wm));
  }
#undef YV4_WG_ISSUE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Domain: 64-channel chunks of input, Cout in whole 16-channel groups (64 .. 2048), 16-bit output, 16-byte aligned
// output / residual views, at most 64 taps; any stride / padding; a scattered output (ys_on) without a residual.
bool conv_wide_h16_applies(const ConvArgsH& a) {
  return (a.Cin & 63) == 0 && a.Kw == a.KH * a.KW * a.Cin && a.KH * a.KW <= 64 && !a.out_f32 && a.Cout >= 64 &&
         (a.Cout & 15) == 0 && a.Cout <= 2048 && ((a.y_cs | a.y_co) & 7) == 0 &&
         (a.res == nullptr || (((a.r_cs | a.r_co) & 7) == 0 && !a.ys_on)) && a.ksplit <= 1 && !(a.ys_on && a.stats);
}

template <bool BF16, int PT, int WAVES_M>
static int launch_wg(const ConvArgsH& a, hipStream_t stream) {
  static LdsAttrOnce once;
  return wide_launch<WideGeom<PT, WAVES_M, false>>(conv_wide_h16_kernel<BF16, PT, WAVES_M>, once, "conv_wide_h16", a, 2, stream);
}

// shape choice and launch: conv_wide_common.h
int conv_wide_h16_pick(const ConvArgsH& a, double* rounds_eff) { return wide_pick(a, false, true, rounds_eff); }

int conv_wide_h16_launch(const ConvArgsH& a, bool bf16, int shape, hipStream_t s) {
  if (shape < 0) shape = conv_wide_h16_pick(a, nullptr);
  if (!wide_shape_fits("conv_wide_h16", false, shape, a.Cout)) return YV4_E_UNSUPPORTED;
#define YV4_WG_CASE(I, PT_, WM_) case I: return bf16 ? launch_wg<true, PT_, WM_>(a, s) : launch_wg<false, PT_, WM_>(a, s);
  switch (shape) {
    YV4_WG_CASE(0, 8, 2)
    YV4_WG_CASE(1, 6, 2)
    YV4_WG_CASE(2, 4, 2)
    YV4_WG_CASE(3, 6, 4)
    YV4_WG_CASE(4, 4, 4)
    default: break;
  }
#undef YV4_WG_CASE
  return YV4_E_INVALID;
}

}  // namespace yv4
