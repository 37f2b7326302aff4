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

namespace yv4 {

typedef float f32x4w __attribute__((ext_vector_type(4)));

template <bool BF16> struct MfmaW;
template <> struct MfmaW<true> {
  static __device__ __forceinline__ f32x4w run(bf16x8 a, bf16x8 b, f32x4w c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaW<false> {
  static __device__ __forceinline__ f32x4w run(f16x8 a, f16x8 b, f32x4w c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int kWgThreads = 512;

template <int PT, int WAVES_M> struct WgGeom {
  static constexpr int WAVES_N = 8 / WAVES_M;
  static constexpr int BN = 64 * WAVES_N;
  static constexpr int WMr = 16 * PT;
  static constexpr int BM = WMr * WAVES_M;
  static constexpr int QA = BM / 64;                // pixel pieces per wave and K tile
  static constexpr int PB = BN / 64;                // weight pieces per wave and K tile
  static constexpr int ABytes = BM * 128;
  static constexpr int BBytes = BN * 128;
  static constexpr int RingBytes = 2 * ABytes + 2 * BBytes;
};

__device__ __forceinline__ int wg_swz_b(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

template <bool BF16, int PT, int WAVES_M>
__global__ __launch_bounds__(kWgThreads, 2) void conv_wide_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef typename Elem<BF16>::V8 V8;
  typedef typename Elem<BF16>::T T;
  typedef WgGeom<PT, WAVES_M> G_;
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
  const int lcB = pc ^ wg_swz_b(srow);
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
      w_rd[ks] = (unsigned)(rw * kRowB + (((fq + 4 * ks) ^ wg_swz_b(rw)) << 4));
    }
  }

  const int nchunks = p.Cin >> 6;
  const int ntaps = p.KH * p.KW;
  const int NK = nchunks * ntaps;            // K tiles per output tile (chunk-major, taps inside)

  const bool has2 = p.s2 != nullptr;
  for (int c = tid; c < p.Cout; c += kWgThreads) {
    aff[c] = p.s1[c];
    aff[p.Cout + c] = p.t1[c];
    aff[2 * p.Cout + c] = has2 ? p.s2[c] : 1.f;
    aff[3 * p.Cout + c] = has2 ? p.t2[c] : 0.f;
  }

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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // (also publishes the affine)

  unsigned T_ = 0u;
  for (int vt = (int)blockIdx.x; vt < ntiles; vt += nwg) {
    const unsigned tile = tile_of(vt);
    const int tile_n = (int)(tile % (unsigned)p.tiles_n);
    const int tile_m = (int)(tile / (unsigned)p.tiles_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    f32x4w acc[PT][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[pt][t] = f32x4w{0.f, 0.f, 0.f, 0.f};

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

    // ---- epilogue (conv3x3_wide_h16.hip's): lane (fr, fq) owns pixel m0 + wm WMr + 16 pt + fr, channels cl .. cl + 15 ----
    const int cl = n0 + wn * 64 + 16 * fq;
    const bool c_ok = cl + 15 < p.Cout;
    const int ca = c_ok ? cl : 0;
    float s1[16], t1[16];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      const float4 a = *reinterpret_cast<const float4*>(aff + ca + u), b = *reinterpret_cast<const float4*>(aff + p.Cout + ca + u);
      s1[u] = a.x; s1[u + 1] = a.y; s1[u + 2] = a.z; s1[u + 3] = a.w;
      t1[u] = b.x; t1[u + 1] = b.y; t1[u + 2] = b.z; t1[u + 3] = b.w;
    }
    float st[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) st[u] = 0.f;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int m = m0 + wm * WMr + 16 * pt + fr;
      const bool ok = c_ok && m < p.M;
      float v[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * t + j] = __builtin_fmaf(acc[pt][t][j], s1[4 * t + j], t1[4 * t + j]);
      {
        float lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = v[e]; hi[e] = v[e + 8]; }
        act_row8(lo, p.act1, p.slope1);
        act_row8(hi, p.act1, p.slope1);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = lo[e]; v[e + 8] = hi[e]; }
      }
      if (p.res && ok) {
        const T* rp = reinterpret_cast<const T*>(p.res) + (int64_t)m * p.r_cs + p.r_co + cl;
        const V8 r0 = *reinterpret_cast<const V8*>(rp), r1 = *reinterpret_cast<const V8*>(rp + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] += (float)r0[e]; v[e + 8] += (float)r1[e]; }
      }
      if (has2) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_fmaf(v[u], aff[2 * p.Cout + ca + u], aff[3 * p.Cout + ca + u]);
        float lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = v[e]; hi[e] = v[e + 8]; }
        act_row8(lo, p.act2, p.slope2);
        act_row8(hi, p.act2, p.slope2);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = lo[e]; v[e + 8] = hi[e]; }
      }
      if (ok) {
        V8 o0, o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { o0[e] = (T)v[e]; o1[e] = (T)v[e + 8]; }
        T* yp = reinterpret_cast<T*>(p.y) + out_row_h(p, m) * p.y_cs + p.y_co + cl;
        *reinterpret_cast<V8*>(yp) = o0;
        *reinterpret_cast<V8*>(yp + 8) = o1;
        if (p.stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float a = (float)o0[e], b = (float)o1[e];
            st[e] += a; st[16 + e] += a * a;
            st[8 + e] += b; st[24 + e] += b * b;
          }
        }
      }
    }
    if (p.stats) {
      int idx = 0;
#pragma unroll
      for (int sft = 0; sft < 4; ++sft) {
        const int half = 16 >> sft;
        const bool bit = (lane >> sft) & 1;
#pragma unroll
        for (int i = 0; i < half; ++i) {
          const float send = bit ? st[i] : st[i + half];
          const float recv = __shfl_xor(send, 1 << sft);
          st[i] = (bit ? st[i + half] : st[i]) + recv;
        }
        idx += bit ? half : 0;
      }
      if (c_ok) {
        const StatRep rep = stat_rep(p.stats, (unsigned)((tile_m * WAVES_M + wm)), p.Cout);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int id = idx + k;
          stat_add(rep, (id >> 4) * p.Cout + cl + (id & 15), st[k]);
        }
      }
    }
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

static int g_wg_cus = 0;
static int wg_cus() {
  if (g_wg_cus == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
    g_wg_cus = cus;
  }
  return g_wg_cus;
}

template <bool BF16, int PT, int WAVES_M>
static int launch_wg(const ConvArgsH& a, hipStream_t stream) {
  typedef WgGeom<PT, WAVES_M> G_;
  ConvArgsH p = a;
  const int tiles_m = (p.M + G_::BM - 1) / G_::BM;
  p.tiles_n = (p.Cout + G_::BN - 1) / G_::BN;
  p.fd_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fastdiv((unsigned)p.Wo);
  const long long tiles = (long long)tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv wide h16: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  const size_t lds = (size_t)G_::RingBytes + (size_t)4 * p.Cout * 4;
  if (lds > 160 * 1024) {
    set_error("conv wide h16: %zu bytes of LDS for this tile shape and Cout", lds);
    return YV4_E_UNSUPPORTED;
  }
  const long long xb = (long long)p.N * p.H * p.W * p.x_cs * 2, wb = (long long)p.Cout * p.Kw * 2;
  auto kern = conv_wide_h16_kernel<BF16, PT, WAVES_M>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024, "conv_wide_h16")) return rc;
  const int cus = wg_cus();
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kWgThreads), lds, stream, p, (unsigned)xb, (unsigned)wb, (int)tiles);
  YV4_CHECK_LAUNCH("conv_wide_h16");
  return YV4_OK;
}

// tile shapes (pixel tiles per wave, waves along M): 256 x 256, 192 x 256, 128 x 256, 384 x 128, 256 x 128
struct WgShape { int pt, wm; };
static const WgShape kWgShapes[5] = {{8, 2}, {6, 2}, {4, 2}, {6, 4}, {4, 4}};
static size_t wg_lds(int pt, int wmv, int Cout) {
  const int bm = 16 * pt * wmv, bn = 64 * (8 / wmv);
  return (size_t)2 * (bm + bn) * 128 + (size_t)16 * Cout;
}
int conv_wide_h16_pick(const ConvArgsH& a, double* rounds_eff) {
  const int cus = wg_cus();
  int best = -1;
  double best_cost = 0.0;
  for (int i = 0; i < 5; ++i) {
    const int pt = kWgShapes[i].pt, wmv = kWgShapes[i].wm;
    const int bm = 16 * pt * wmv, bn = 64 * (8 / wmv);
    if (wg_lds(pt, wmv, a.Cout) > 160 * 1024) continue;
    if (bn > ((a.Cout + 127) / 128) * 128) continue;
    const long long tiles = ((long long)a.M + bm - 1) / bm * ((a.Cout + bn - 1) / bn);
    const long long rounds = (tiles + cus - 1) / cus;
    const double eff = pt == 8 ? 1.0 : (pt == 6 ? 1.04 : 1.12);
    const double cost = (double)rounds * bm * bn * eff;
    if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  if (rounds_eff && best >= 0) *rounds_eff = best_cost / ((double)a.M * a.Cout / cus);
  return best;
}

int conv_wide_h16_launch(const ConvArgsH& a, bool bf16, int shape, hipStream_t s) {
  if (shape < 0) shape = conv_wide_h16_pick(a, nullptr);
  if (shape < 0 || shape >= 5 || wg_lds(kWgShapes[shape].pt, kWgShapes[shape].wm, a.Cout) > 160 * 1024) {
    set_error("conv wide h16: no tile shape of this layer fits the LDS");
    return YV4_E_UNSUPPORTED;
  }
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
