// Fused implicit-GEMM convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fp32 -- there is no xf32 on CDNA4).
//
// Replaces, for inference, the reference's  Conv2d -> BatchNorm2d(eval) -> Mish
// (mmcv ConvModule as instantiated by mmdet/models/backbones/darknetcsp.py:15-35),
// the Bottleneck residual add (darknetcsp.py:60-64), the CSP-level
// cat -> BN -> act (darknetcsp.py:106-109,149-153,220-229) and the biased head
// conv (mmdet/models/dense_heads/yolocsp_head.py:180-185, 216-220).
//
// GEMM view:  M = N*Ho*Wo output pixels, Ncol = Cout, K = KH*KW*Cin.
//   A[m][k]  = x[n, ho*s-p+kh, wo*s-p+kw, ci]    (NHWC gather, zero outside)
//   B[k][co] = w[co][k]                          (weights stored K-contiguous)
// A workgroup (256 threads = 4 waves) owns a BM x BN output tile; the K loop
// stages BK=32 deep slices of A and B through double-buffered LDS (rows padded to
// 36 floats so the ds_read_b128 fragment reads are bank-conflict free), global
// loads for slice t+1 are in flight while slice t feeds the MFMAs.
// Per MFMA the lane map is  A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31],
// D[row=(reg&3)+8*(reg>>2)+4*(lane>>5)][col=lane&31]; rows are pixels and columns
// are output channels, so every accumulator register stores 2 x 128 contiguous
// bytes of NHWC output.
#include "conv_f32_common.h"

namespace yv4 {

// (ConvArgs, out_row, lds_dma16, make_rsrc: conv_f32_common.h -- shared with conv3x3_wide_f32.hip)

template <int BM, int BN, int WAVES_M, int WAVES_N, bool UNIFORM_TAP>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_f32_kernel(ConvArgs p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int TM = BM / WAVES_M / 32;  // 32x32 MFMA tiles per wave along M
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int PA = BM / 32;            // staging passes (32 rows per pass)
  constexpr int PB = BN / 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                   // [2][BM][kLDK]
  float* Bs = smem + 2 * BM * kLDK;   // [2][BN][kLDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int r = lane & 31;
  const int h = lane >> 5;

  const int tile_n = blockIdx.x % p.tiles_n;
  const int tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- staging coordinates: thread -> (row = tid/8 + 32*pass, 16-byte chunk cc) ----
  const int cc = tid & 7;
  const int srow = tid >> 3;

  const float* a_ptr[PA];  // &x[n, ho*s-p, wo*s-p, x_co + cc*4]  (may be out of range)
  int a_hi0[PA], a_wi0[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int m = m0 + srow + 32 * q;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad;
      const int wi0 = wo * p.stride - p.pad;
      a_hi0[q] = hi0;
      a_wi0[q] = wi0;
      a_ptr[q] = p.x + ((int64_t)(n * p.H + hi0) * p.W + wi0) * p.x_cs + p.x_co + cc * 4;
    } else {
      a_hi0[q] = -(1 << 28);  // never in range
      a_wi0[q] = -(1 << 28);
      a_ptr[q] = p.x;
    }
  }
  const float* b_ptr[PB];
  bool b_ok[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int co = n0 + srow + 32 * q;
    b_ok[q] = co < p.Cout;
    b_ptr[q] = p.w + (int64_t)(b_ok[q] ? co : 0) * p.Kw + cc * 4;
  }

  float4 ra[PA], rb[PB];

  auto load_slice = [&](int kt) {
    const int kbase = kt * kBK;
    if (UNIFORM_TAP) {
      // Cin % 32 == 0: the whole slice lies in one (kh,kw) tap.
      const int tap = kbase / p.Cin;
      const int c0 = kbase - tap * p.Cin;
      const int kh = tap / p.KW;
      const int kw = tap - kh * p.KW;
      const int64_t step = ((int64_t)kh * p.W + kw) * p.x_cs + c0;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int hi = a_hi0[q] + kh;
        const int wi = a_wi0[q] + kw;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        ra[q] = ok ? *reinterpret_cast<const float4*>(a_ptr[q] + step)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        rb[q] = b_ok[q] ? *reinterpret_cast<const float4*>(b_ptr[q] + kbase)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      // generic: Cin % 4 == 0, so a 4-float chunk never straddles a tap.
      const int k0 = kbase + cc * 4;
      const bool kok = k0 < p.K;
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int kh = tap / p.KW;
      const int kw = tap - kh * p.KW;
      const int64_t step = ((int64_t)kh * p.W + kw) * p.x_cs + c0 - cc * 4;
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int hi = a_hi0[q] + kh;
        const int wi = a_wi0[q] + kw;
        const bool ok = kok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        ra[q] = ok ? *reinterpret_cast<const float4*>(a_ptr[q] + step)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        rb[q] = (b_ok[q] && kok) ? *reinterpret_cast<const float4*>(b_ptr[q] + kbase)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  auto store_slice = [&](int buf) {
    float* as = As + buf * BM * kLDK;
    float* bs = Bs + buf * BN * kLDK;
#pragma unroll
    for (int q = 0; q < PA; ++q)
      *reinterpret_cast<float4*>(as + (srow + 32 * q) * kLDK + cc * 4) = ra[q];
#pragma unroll
    for (int q = 0; q < PB; ++q)
      *reinterpret_cast<float4*>(bs + (srow + 32 * q) * kLDK + cc * 4) = rb[q];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + kBK - 1) / kBK;

  load_slice(0);
  store_slice(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_slice(kt + 1);

    const float* as = As + buf * BM * kLDK + (wm * TM * 32 + r) * kLDK + 4 * h;
    const float* bs = Bs + buf * BN * kLDK + (wn * TN * 32 + r) * kLDK + 4 * h;
#pragma unroll
    for (int j = 0; j < kBK / 8; ++j) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const float4*>(as + i * 32 * kLDK + 8 * j);
#pragma unroll
      for (int i = 0; i < TN; ++i)
        fb[i] = *reinterpret_cast<const float4*>(bs + i * 32 * kLDK + 8 * j);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[jn].x, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[jn].y, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[jn].z, acc[i][jn], 0, 0, 0);
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[jn].w, acc[i][jn], 0, 0, 0);
        }
    }

    if (kt + 1 < nk) store_slice(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: affine1 -> act1 -> (+residual) -> [affine2 -> act2] -> NHWC store ----
  const bool has2 = p.s2 != nullptr;
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int co = n0 + wn * TN * 32 + jn * 32 + r;
    const bool cok = co < p.Cout;
    const float s1 = cok ? p.s1[co] : 0.f;
    const float t1 = cok ? p.t1[co] : 0.f;
    const float s2 = (cok && has2) ? p.s2[co] : 1.f;
    const float t2 = (cok && has2) ? p.t2[co] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (cok && m < p.M) {
          float v = acc[i][jn][e] * s1 + t1;
          v = apply_act(v, p.act1, p.slope1);
          if (p.res) v += p.res[(int64_t)m * p.r_cs + p.r_co + co];
          if (has2) v = apply_act(v * s2 + t2, p.act2, p.slope2);
          p.y[(int64_t)m * p.y_cs + p.y_co + co] = v;
        }
      }
    }
  }
}


// Epilogue of one 32x32 accumulator tile through a wave-private LDS patch: the MFMA result has
// the output channel on the lane and 16 pixels in registers, which stores as 16 dword
// instructions touching 2 x 128 B each; transposed through LDS every lane owns 4 consecutive
// channels of one pixel, so the tile leaves (and the residual arrives) in 4 dwordx4
// instructions per lane and the per-channel scale/shift become vector loads.
// `vec_ok` (uniform): all views are 16-byte aligned per pixel (false for the 255-channel heads).
// the activation of a lane's 4 values behind ONE uniform branch on the activation id (apply_act() per element left
// a scalar compare-and-branch chain per element: the compiler does not unswitch the unrolled loop)
__device__ __forceinline__ void act_row4(float4& v, int act, float slope) {
  switch (act) {
    case YV4_ACT_MISH:
      v.x = apply_act(v.x, YV4_ACT_MISH, 0.f); v.y = apply_act(v.y, YV4_ACT_MISH, 0.f);
      v.z = apply_act(v.z, YV4_ACT_MISH, 0.f); v.w = apply_act(v.w, YV4_ACT_MISH, 0.f);
      break;
    case YV4_ACT_LEAKY:
      v.x = v.x >= 0.f ? v.x : v.x * slope; v.y = v.y >= 0.f ? v.y : v.y * slope;
      v.z = v.z >= 0.f ? v.z : v.z * slope; v.w = v.w >= 0.f ? v.w : v.w * slope;
      break;
    case YV4_ACT_SWISH:
      v.x = apply_act(v.x, YV4_ACT_SWISH, 0.f); v.y = apply_act(v.y, YV4_ACT_SWISH, 0.f);
      v.z = apply_act(v.z, YV4_ACT_SWISH, 0.f); v.w = apply_act(v.w, YV4_ACT_SWISH, 0.f);
      break;
    default:
      break;
  }
}

__device__ __forceinline__ void epilogue_tile(const ConvArgs& p, const f32x16& acc, float* ep, int lane, int m_base,
                                              int co_base, bool vec_ok, bool has2) {
  const int r = lane & 31, h = lane >> 5;
  constexpr int kPitch = 36;
#pragma unroll
  for (int e = 0; e < 16; ++e) ep[((e & 3) + 8 * (e >> 2) + 4 * h) * kPitch + r] = acc[e];
  const int c4 = (lane & 7) * 4;
  const int co = co_base + c4;
  if (p.ksplit > 1) {                       // split-K partial: raw accumulators, dense [M][ws_cs] slab (ws_cs % 4 == 0)
    if (co < p.ws_cs) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = (lane >> 3) + 8 * k;
        const int m = m_base + row;
        if (m < p.M)
          *reinterpret_cast<float4*>(p.y + (int64_t)m * p.ws_cs + co) = *reinterpret_cast<const float4*>(ep + row * kPitch + c4);
      }
    }
    return;
  }
  if (vec_ok && co + 3 < p.Cout) {
    const float4 s1 = *reinterpret_cast<const float4*>(p.s1 + co);
    const float4 t1 = *reinterpret_cast<const float4*>(p.t1 + co);
    float4 s2 = make_float4(1.f, 1.f, 1.f, 1.f), t2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has2) {
      s2 = *reinterpret_cast<const float4*>(p.s2 + co);
      t2 = *reinterpret_cast<const float4*>(p.t2 + co);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = (lane >> 3) + 8 * k;
      const int m = m_base + row;
      const float4 a = *reinterpret_cast<const float4*>(ep + row * kPitch + c4);
      if (m < p.M) {
        float4 v = make_float4(a.x * s1.x + t1.x, a.y * s1.y + t1.y, a.z * s1.z + t1.z, a.w * s1.w + t1.w);
        act_row4(v, p.act1, p.slope1);
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + (int64_t)m * p.r_cs + p.r_co + co);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (has2) {
          v = make_float4(v.x * s2.x + t2.x, v.y * s2.y + t2.y, v.z * s2.z + t2.z, v.w * s2.w + t2.w);
          act_row4(v, p.act2, p.slope2);
        }
        *reinterpret_cast<float4*>(p.y + out_row(p, m) * p.y_cs + p.y_co + co) = v;
      }
    }
  } else {
    // rows that are not 16-byte aligned (the 255-channel pred maps) or a ragged column tile: lane (r, h) takes channel
    // co_base + r of the rows 2k + h, so a store instruction writes two runs of 32 consecutive floats (coalesced
    // whatever the row's alignment).  Same arithmetic, element for element, as the vector path.
    const int c = co_base + r;
    if (c < p.Cout) {
      const float sc1 = p.s1[c], sh1 = p.t1[c];
      const float sc2 = has2 ? p.s2[c] : 1.f, sh2 = has2 ? p.t2[c] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = 2 * k + h;
        const int m = m_base + row;
        if (m >= p.M) continue;
        float v = ep[row * kPitch + r] * sc1 + sh1;
        v = apply_act(v, p.act1, p.slope1);
        if (p.res) v += p.res[(int64_t)m * p.r_cs + p.r_co + c];
        if (has2) v = apply_act(v * sc2 + sh2, p.act2, p.slope2);
        p.y[out_row(p, m) * p.y_cs + p.y_co + c] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Fast path (Cin % 32 == 0): LDS-DMA staging.  The next K slice goes global -> LDS directly
// (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, no VGPR round trip and no
// ds_write: in the register-staged kernel above the VGPR->LDS store path is the largest
// non-MFMA cost, ~12 % in an ablation).
// An LDS-DMA instruction writes lane l at base + 16*l, so the LDS image is unpadded
// [row][32 floats] with 128-byte rows; ds_read_b128 conflicts are avoided by an XOR
// swizzle of the 16-byte chunk index with (row>>1)&7, applied on the SOURCE side (which
// global chunk a lane fetches) and on the read side.  Out-of-range buffer offsets deliver
// zeros, which implements conv padding, the M tail and the Cout tail.
// ---------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int NBUF>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_f32_dma_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes) {
  static_assert(NBUF >= 2 && NBUF <= 4, "ring of 2..4 slices");
  constexpr int D = NBUF - 1;           // prefetch distance in K slices
  constexpr int kDmaPerSlice = BM / 32 + BN / 32;   // LDS-DMA instructions per wave per slice
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int PA = BM / 32;   // DMA passes over A: 32 rows per pass (8 rows per wave-instruction)
  constexpr int PB = BN / 32;
  constexpr int kRow = kBK;     // floats per LDS row (128 B, unpadded)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [NBUF][BM][32]
  float* Bs = smem + NBUF * BM * kRow;   // [NBUF][BN][32]

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
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const unsigned tile = tile_s / (unsigned)ksplit;          // the splits of one tile are neighbours (same L2)
  const int split = (int)(tile_s - tile * (unsigned)ksplit);
  const int tile_n = tile % p.tiles_n;
  const int tile_m = tile / p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  if (ksplit > 1) p.y = p.ws + (size_t)split * p.M * p.ws_cs;

  const u32x4_t rsA = make_rsrc(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem;   // LDS byte address of the carve

  // staging: this lane fills LDS row (32*q + 8*wave + lane/8), physical chunk lane%8
  const int srow = 8 * wave + (lane >> 3);
  const int pc = lane & 7;
  unsigned a_off[PA];
  unsigned long long a_mask[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int row = srow + 32 * q;
    const int lc = pc ^ ((row >> 1) & 7);    // logical chunk this lane must fetch
    const int m = m0 + row;
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
      off = (unsigned)((((int64_t)(n * p.H + hi0) * p.W + wi0) * p.x_cs + p.x_co + lc * 4) * 4);
      mk = tap_mask(hi0, wi0, p.KH, p.KW, p.H, p.W);
    }
    a_off[q] = off;
    a_mask[q] = mk;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int row = srow + 32 * q;
    const int lc = pc ^ ((row >> 1) & 7);
    const int co = n0 + row;
    b_off[q] = co < p.Cout ? (unsigned)(((int64_t)co * p.Kw + lc * 4) * 4) : kOOB;
  }

  // K is walked channel-chunk-major: the KH*KW taps of one 32-channel chunk in consecutive slices, then the next
  // chunk.  A tap re-reads the 128-B lines its neighbours read one (kw) or KW (kh) slices ago, which are still in the
  // XCD's L2; with the taps outermost each line came back Cin/32 slices later, after every concurrent tile had streamed
  // its whole window through the 4 MB L2 (profiles/r01_layer_traffic.md).  The weights stay [Cout][kh][kw][cin].
  const int ntaps = p.KH * p.KW;
  int s_tap = 0, s_c0 = 0, s_kh = 0, s_kw = 0;
  unsigned s_kb = 0;
  const int slice0 = ksplit > 1 ? split * p.ks_slices : 0;      // first K slice of this workgroup
  if (slice0) {
    const int chunk = fd_div(slice0, p.fd_taps);
    s_tap = slice0 - chunk * ntaps;
    s_c0 = chunk * kBK;
    s_kh = fd_div(s_tap, p.fd_kw);
    s_kw = s_tap - s_kh * p.KW;
    s_kb = (unsigned)((s_tap * p.Cin + s_c0) * 4);
  }

  // The kDmaPerSlice LDS-DMA pieces of a slice: the prologue issues them in one block (YV4_V3_DMA); in the K loop they
  // go out between the MFMAs of the slice being computed, a quarter per 8-deep K step (YV4_V3_DMA_PIECES inside
  // YV4_V3_COMPUTE) -- a piece costs 60-180 cycles of issue (MI355X_MICROARCH.md) during which the wave cannot feed the
  // matrix pipe unless its own MFMAs are already in flight (the weight-stationary kernel's stamps).
  unsigned d_step = 0u, d_la = 0u, d_lb = 0u;
#define YV4_V3_DMA_SETUP(BUF)                                                       \
  {                                                                                 \
    d_step = (unsigned)((((int64_t)s_kh * p.W + s_kw) * p.x_cs + s_c0) * 4);        \
    d_la = lds_base + (unsigned)(((BUF) * BM + 8 * wave) * kRow * 4);               \
    d_lb = lds_base + (unsigned)((NBUF * BM + (BUF) * BN + 8 * wave) * kRow * 4);   \
  }
#define YV4_V3_DMA_PIECE(Q)                                                         \
  {                                                                                 \
    if ((Q) < PA) {                                                                 \
      const int qa_ = (Q) < PA ? (Q) : 0;                                           \
      const bool ok = (a_mask[qa_] >> s_tap) & 1ull;                                \
      lds_dma16(rsA, d_la + 32 * qa_ * kRow * 4, ok ? a_off[qa_] + d_step : kOOB, 0u); \
    } else {                                                                        \
      const int qb_ = (Q) >= PA ? (Q) - PA : 0;                                     \
      lds_dma16(rsB, d_lb + 32 * qb_ * kRow * 4, b_off[qb_], s_kb);                  \
    }                                                                               \
  }
#define YV4_V3_DMA_ADVANCE()                                                        \
  {                                                                                 \
    s_tap += 1;                                                                     \
    s_kw += 1;                                                                      \
    const int wrap_w = s_kw == p.KW ? 1 : 0;                                        \
    s_kw = wrap_w ? 0 : s_kw;                                                       \
    s_kh += wrap_w;                                                                 \
    const int wrap_t = s_tap == ntaps ? 1 : 0;                                      \
    s_tap = wrap_t ? 0 : s_tap;                                                     \
    s_kh = wrap_t ? 0 : s_kh;                                                       \
    s_c0 += wrap_t ? kBK : 0;                                                       \
    s_kb = (unsigned)((s_tap * p.Cin + s_c0) * 4);                                  \
  }
#define YV4_V3_DMA(BUF)                                                             \
  {                                                                                 \
    YV4_V3_DMA_SETUP(BUF);                                                          \
    _Pragma("unroll") for (int q = 0; q < kDmaPerSlice; ++q) YV4_V3_DMA_PIECE(q);   \
    YV4_V3_DMA_ADVANCE();                                                           \
  }
  constexpr bool kPiecewise = TM * TN >= 2;   // (the 64 x 64 tile has ONE MFMA quad per K step to issue under: no gain, -2 %)
#define YV4_V3_DMA_PIECES(J)                                                        \
  if (kPiecewise && dma_on) {                                                       \
    __builtin_amdgcn_sched_barrier(0);                                              \
    _Pragma("unroll") for (int q = (J) * kDmaPerSlice / 4; q < ((J) + 1) * kDmaPerSlice / 4; ++q) YV4_V3_DMA_PIECE(q); \
    __builtin_amdgcn_sched_barrier(0);                                              \
  }

  // fragment read addresses: row*128 B + ((chunk ^ swz) << 4); chunk = 2j + h
  unsigned a_rd[TM], b_rd[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * TM * 32 + i * 32 + r;
    a_rd[i] = (unsigned)(row * kRow * 4 + ((((row >> 1) & 7) ^ h) << 4));
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * TN * 32 + i * 32 + r;
    b_rd[i] = (unsigned)(row * kRow * 4 + ((((row >> 1) & 7) ^ h) << 4));
  }

#define YV4_V3_COMPUTE(BUF)                                                         \
  {                                                                                 \
    const char* as_ = reinterpret_cast<const char*>(As + (BUF) * BM * kRow);        \
    const char* bs_ = reinterpret_cast<const char*>(Bs + (BUF) * BN * kRow);        \
    _Pragma("unroll") for (int j = 0; j < kBK / 8; ++j) {                           \
      float4 fa[TM], fb[TN];                                                        \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                \
          fa[i] = *reinterpret_cast<const float4*>(as_ + (a_rd[i] ^ (unsigned)(j << 5))); \
      _Pragma("unroll") for (int i = 0; i < TN; ++i)                                \
          fb[i] = *reinterpret_cast<const float4*>(bs_ + (b_rd[i] ^ (unsigned)(j << 5))); \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                \
        _Pragma("unroll") for (int jn = 0; jn < TN; ++jn) {                         \
          f32x16& ac_ = (j & 1) ? acc2[i][jn] : acc[i][jn];                         \
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[jn].x, ac_, 0, 0, 0); \
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[jn].y, ac_, 0, 0, 0); \
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[jn].z, ac_, 0, 0, 0); \
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[jn].w, ac_, 0, 0, 0); \
          if (i == 0 && jn == 0) YV4_V3_DMA_PIECES(j)                               \
        }                                                                           \
    }                                                                               \
  }

  // Two accumulator sets, alternating every 8 K values: each fp32 chain is half as long (the
  // sequential K = 4608 chain of a single set measured 1.7x the CPU reference's rounding error
  // against float64 at full depth, tests/test_gpu_fullsize.py); summed once before the epilogue.
  f32x16 acc[TM][TN], acc2[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

  // wait until at most `newer` later slices of this wave's DMAs are still in flight
#define YV4_V3_WAIT(NEWER)                                                          \
  {                                                                                 \
    if ((NEWER) >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * kDmaPerSlice) : "memory"); \
    else if ((NEWER) == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * kDmaPerSlice) : "memory"); \
    else if ((NEWER) == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(1 * kDmaPerSlice) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
  }

  const int nk_all = p.K / kBK;
  const int nk = ksplit > 1 ? (nk_all - slice0 < p.ks_slices ? nk_all - slice0 : p.ks_slices) : nk_all;
  int issued = 0;        // slices whose DMA has been issued
  int wbuf = 0;          // ring slot the next DMA goes to
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < nk) {
      YV4_V3_DMA(wbuf);
      ++issued;
      wbuf = wbuf + 1 == NBUF ? 0 : wbuf + 1;
    }
  }
  YV4_V3_WAIT(issued - 1);             // slice 0 landed (this wave's part) ...
  __builtin_amdgcn_s_barrier();        // ... and everybody else's

  int rbuf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool dma_on = issued < nk;   // slot wbuf was last read before the previous barrier
    if (dma_on) {
      YV4_V3_DMA_SETUP(wbuf);
      if (!kPiecewise) {
#pragma unroll
        for (int q = 0; q < kDmaPerSlice; ++q) YV4_V3_DMA_PIECE(q);
      }
    }
    __builtin_amdgcn_s_setprio(1);   // keep the MFMA cluster ahead of the other resident waves' setup code
    YV4_V3_COMPUTE(rbuf);
    __builtin_amdgcn_s_setprio(0);
    if (dma_on) {
      YV4_V3_DMA_ADVANCE();
      ++issued;
      wbuf = wbuf + 1 == NBUF ? 0 : wbuf + 1;
    }
    rbuf = rbuf + 1 == NBUF ? 0 : rbuf + 1;
    if (kt + 1 < nk) {
      // fragment reads of this slot returned (lgkmcnt(0)), slice kt+1 landed (counted
      // vmcnt: slices kt+2.. may stay in flight), then the workgroup moves on.
      YV4_V3_WAIT(issued - (kt + 2));
      __builtin_amdgcn_s_barrier();
    }
  }
#undef YV4_V3_WAIT
#undef YV4_V3_DMA
#undef YV4_V3_DMA_SETUP
#undef YV4_V3_DMA_PIECE
#undef YV4_V3_DMA_PIECES
#undef YV4_V3_DMA_ADVANCE
#undef YV4_V3_COMPUTE

  // ---- epilogue: transposed through a wave-private LDS patch (the K-loop buffers are free
  // once every wave has issued its last fragment reads: one more barrier) ----
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] += acc2[i][j][e];
  // BatchNorm statistics of the tile while it is in registers (identity-epilogue training convs; see the same
  // block in conv_mfma_h16.hip)
  if (p.stats) {
    const StatRep rep = stat_rep(p.stats, (unsigned)(tile_m), p.Cout);
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float su = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * TM * 32 + i * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc[i][jn][e];
          if (mb + (e & 3) + 8 * (e >> 2) < p.M) { su += v; sq += v * v; }
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
  const bool vec_ok = ((p.y_cs | p.y_co) & 3) == 0 && (p.res == nullptr || ((p.r_cs | p.r_co) & 3) == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* ep = smem + wave * (32 * 36);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
      epilogue_tile(p, acc[i][jn], ep, lane, m0 + wm * TM * 32 + i * 32, n0 + wn * TN * 32 + jn * 32, vec_ok, has2);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NBUF>
static int launch_conv_dma(const ConvArgs& a, hipStream_t stream) {
  constexpr size_t lds = (size_t)NBUF * (BM + BN) * kBK * sizeof(float);
  ConvArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  p.fd_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fastdiv((unsigned)p.Wo);
  p.fd_taps = make_fastdiv((unsigned)(p.KH * p.KW));
  p.fd_kw = make_fastdiv((unsigned)p.KW);
  const long long tiles = (long long)tiles_m * p.tiles_n * (p.ksplit > 1 ? p.ksplit : 1);
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  const long long xb = (long long)p.N * p.H * p.W * p.x_cs * 4, wb = (long long)p.Cout * p.Kw * 4;
  if (xb >= 0xFFFFFFF0LL || wb >= 0xFFFFFFF0LL) {
    set_error("conv dma: tensors of 4 GiB or more are not addressable through a buffer descriptor");
    return YV4_E_UNSUPPORTED;
  }
  auto kern = conv_mfma_f32_dma_kernel<BM, BN, WAVES_M, WAVES_N, NBUF>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), lds, "conv_mfma_f32_dma")) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(kThreads), lds, stream, p, (unsigned)xb, (unsigned)wb);
  YV4_CHECK_LAUNCH("conv_mfma_f32_dma");
  return YV4_OK;
}


// ---------------------------------------------------------------------------------
// Stem: 3x3 / stride 1 / pad 1 convolution of the 3-channel image (stored NHWC with C padded to
// 4), Cout <= 64.  K = 9 taps x 4 channels = 36 is too shallow for the LDS-staged kernels (they
// spend their time in prologue/epilogue) and the layer is bound by its own OUTPUT
// (N*H*W*Cout*4 B = 1.5 GB at batch 32, 608^2, Cout 32), so this kernel keeps everything in
// registers: a wave owns 32 consecutive pixels of one image row x 32 output channels
// (one 32x32 MFMA tile), fetches its 9 x 8-byte input taps straight into the MFMA A operand
// (lane (r,h): pixel r, channels 2h,2h+1 of each tap; out-of-image taps come back as zeros from
// the buffer descriptor), holds the 18 weight values it needs for the whole kernel, issues
// 18 MFMAs per tile and streams the epilogue to HBM in 128-byte rows.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void act_row16(float (&v)[16], int act, float slope) {
  switch (act) {
    case YV4_ACT_MISH:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = apply_act(v[e], YV4_ACT_MISH, 0.f);
      break;
    case YV4_ACT_LEAKY:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope;
      break;
    case YV4_ACT_SWISH:
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = apply_act(v[e], YV4_ACT_SWISH, 0.f);
      break;
    default:
      break;
  }
}

// one value to y[row_base + lane_off]: row_base (elements) is wave-uniform, lane_off a 32-bit per-lane constant
template <int OUT>
__device__ __forceinline__ void stem_store(float* y, long long row_base, unsigned lane_off, float v) {
  if (OUT == 0) (y + row_base)[lane_off] = v;
  else if (OUT == 1) (reinterpret_cast<_Float16*>(y) + row_base)[lane_off] = (_Float16)v;
  else (reinterpret_cast<__bf16*>(y) + row_base)[lane_off] = (__bf16)v;
}

// OUT: 0 fp32 (p.y), 1 fp16, 2 bf16 (p.y reinterpreted; the 16-bit inference path keeps the image
// and this layer's arithmetic in fp32 and only rounds the layer's output).
template <int TN, int OUT>
__global__ __launch_bounds__(kThreads) void conv_stem3x3_kernel(ConvArgs p, unsigned x_bytes, int tiles_w, long long ntiles) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int r = lane & 31;
  const int h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, x_bytes, 0x00020000);
  constexpr unsigned kOOB = 0xFFFFFFF0u;

  // weights: B[k][cout r] with k = (tap, ci = 2h + j)
  float wv[TN][9][2];
  float s1[TN], t1[TN];
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int co = jn * 32 + r;
    const bool cok = co < p.Cout;
    s1[jn] = cok ? p.s1[co] : 0.f;
    t1[jn] = cok ? p.t1[co] : 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) wv[jn][tap][j] = cok ? p.w[(size_t)co * p.Kw + tap * 4 + 2 * h + j] : 0.f;
  }

  // tile walk in wave-uniform 32-bit arithmetic with magic-number divisors (the 64-bit per-lane divides of the
  // first version were a third of the instructions of a tile)
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: XCD c walks the c-th eighth of the tile
  // list with all its waves side by side, so the rows a tile shares with the tiles above and below it (3 input rows
  // per output row) are fetched by one L2 once instead of by three (measured 3.4x the input before, FETCH_SIZE).
  const int nblk_all = (int)gridDim.x;
  const int nch = nblk_all < 8 ? nblk_all : 8;
  const int chunk = (int)blockIdx.x % nch, local = (int)blockIdx.x / nch;
  const int nblk = (nblk_all - chunk + nch - 1) / nch;               // workgroups walking this chunk
  const int cq = (int)ntiles / nch, crem = (int)ntiles % nch;
  const int t_lo = chunk * cq + (chunk < crem ? chunk : crem);
  const int t_hi = t_lo + cq + (chunk < crem ? 1 : 0);
  const int wave_id = __builtin_amdgcn_readfirstlane(local * 4 + (tid >> 6));
  const int nwaves = nblk * 4;
  for (int t = t_lo + wave_id; t < t_hi; t += nwaves) {
    const long long ty = fd_div(t, p.fd_wo);     // n*H + y   (fd_wo: tiles per row, fd_hw: H -- set by launch_conv_stem)
    const int tx = t - (int)ty * tiles_w;
    const int y = (int)ty - fd_div((int)ty, p.fd_hw) * p.H;
    const int x = tx * 32 + r;
    // byte offset of x[n, y, x, x_co + 2h]
    const unsigned base = (unsigned)(((ty * p.W + x) * p.x_cs + p.x_co + 2 * h) * 4);
    u32x2 a[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const bool rok = (unsigned)(y + kh - 1) < (unsigned)p.H;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const bool ok = rok && (unsigned)(x + kw - 1) < (unsigned)p.W;
        const unsigned off = base + (unsigned)((((kh - 1) * p.W + (kw - 1)) * p.x_cs) * 4);
        a[kh * 3 + kw] = __builtin_amdgcn_raw_buffer_load_b64(rsA, ok ? off : kOOB, 0, 0);
      }
    }
    f32x16 acc[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[jn][e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[tap].x), wv[jn][tap][0], acc[jn], 0, 0, 0);
        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[tap].y), wv[jn][tap][1], acc[jn], 0, 0, 0);
      }
    // epilogue: lane (r, h) holds channel jn*32 + r of the tile's pixels (e&3) + 8*(e>>2) + 4h.  The row address is
    // wave-uniform (scalar base + one per-lane offset that never changes), the activation sits behind one uniform
    // switch and the bounds test is per tile: the per-element form of all three cost 2.7x the 18 MFMAs in VALU time.
    const long long mrow = ty * p.W + tx * 32;
    const bool full = tx * 32 + 32 <= p.W;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      if (jn * 32 + r >= p.Cout) continue;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[jn][e] * s1[jn] + t1[jn];
      act_row16(v, p.act1, p.slope1);
      const unsigned lane_off = (unsigned)(4 * h * p.y_cs + jn * 32 + r);
      if (full) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          stem_store<OUT>(p.y, (mrow + (e & 3) + 8 * (e >> 2)) * p.y_cs + p.y_co, lane_off, v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (tx * 32 + (e & 3) + 8 * (e >> 2) + 4 * h < p.W)
            stem_store<OUT>(p.y, (mrow + (e & 3) + 8 * (e >> 2)) * p.y_cs + p.y_co, lane_off, v[e]);
      }
    }
  }
}

template <int OUT>
static int launch_conv_stem(const ConvArgs& a, hipStream_t stream) {
  const long long xb = (long long)a.N * a.H * a.W * a.x_cs * 4;
  if (xb >= 0xFFFFFFF0LL) {
    set_error("conv stem: input of 4 GiB or more is not addressable through a buffer descriptor");
    return YV4_E_UNSUPPORTED;
  }
  const int tiles_w = (a.W + 31) / 32;
  const long long ntiles = (long long)a.N * a.H * tiles_w;
  if (ntiles >= (1LL << 31)) {
    set_error("conv stem: %lld tiles do not fit 31 bits", ntiles);
    return YV4_E_UNSUPPORTED;
  }
  long long blocks = (ntiles + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;   // 8 workgroups per CU, grid-stride over the tiles
  ConvArgs p = a;
  p.fd_wo = make_fastdiv((unsigned)tiles_w);   // the stem kernel's tile walk: t / tiles_w, (n*H + y) / H
  p.fd_hw = make_fastdiv((unsigned)a.H);
  if (a.Cout <= 32)
    hipLaunchKernelGGL((conv_stem3x3_kernel<1, OUT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, (unsigned)xb,
                       tiles_w, ntiles);
  else
    hipLaunchKernelGGL((conv_stem3x3_kernel<2, OUT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, (unsigned)xb,
                       tiles_w, ntiles);
  YV4_CHECK_LAUNCH("conv_stem3x3");
  return YV4_OK;
}


template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_conv(const ConvArgs& a, bool uniform_tap, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * kLDK * sizeof(float);
  ConvArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  p.fd_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fastdiv((unsigned)p.Wo);
  const long long tiles = (long long)tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  auto kern_u = conv_mfma_f32_kernel<BM, BN, WAVES_M, WAVES_N, true>;
  auto kern_g = conv_mfma_f32_kernel<BM, BN, WAVES_M, WAVES_N, false>;
  static LdsAttrOnce once_u, once_g;
  if (int rc = ensure_dyn_lds(once_u, reinterpret_cast<const void*>(kern_u), lds, "conv_mfma_f32")) return rc;
  if (int rc = ensure_dyn_lds(once_g, reinterpret_cast<const void*>(kern_g), lds, "conv_mfma_f32")) return rc;
  if (uniform_tap)
    hipLaunchKernelGGL(kern_u, dim3((unsigned)tiles), dim3(kThreads), lds, stream, p);
  else
    hipLaunchKernelGGL(kern_g, dim3((unsigned)tiles), dim3(kThreads), lds, stream, p);
  YV4_CHECK_LAUNCH("conv_mfma_f32");
  return YV4_OK;
}

// ---- 1x1 / stride 1, weight-stationary and persistent (the fp32 form of conv1x1_ws_h16.hip) ---------------------------
// The pointwise layers with Cin <= 256 run at 50-57 % of the fp32 matrix peak on the tiles above (profiles/
// r02_layers.json): a tile's K loop is 2-8 slices, so a workgroup's life is mostly its first-slice latency and its
// epilogue.  Here one 8-wave workgroup per CU keeps its weight slab (BN x Cin floats, <= 64 KB) in LDS for the whole
// layer and every wave walks its own strips of 32 pixels with a private 3-stage LDS-DMA ring (stage = 32 pixels x 32
// channels) that runs on across strips; no barrier after the slab has landed.  The summation order of an output is
// EXACTLY the tile kernels' (slices of 32 channels in order; inside a slice the MFMA K pairs (8j+i, 8j+4+i), i = 0..3;
// two accumulator sets alternating with j, added once at the end), so a layer gives the same bits whichever kernel a
// batch size selects -- the plans' cross-batch bit-exactness (bench.py's output check) holds.
constexpr int kWsfWaves = 8;
constexpr int kWsfThreads = kWsfWaves * 64;
constexpr int kWsfStages = 3;
constexpr int kWsfStageBytes = 4096;   // 32 pixels x 32 channels x 4 bytes
constexpr int kWsfGrid = 256;

template <int NT>
__global__ __launch_bounds__(kWsfThreads, 1) void conv1x1_ws_f32_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes, int ncol,
                                                                        int nstrips, int cpr_shift) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int BN = NT * 32;
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* smem_c = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;
#ifdef YV4_MEASURE
  // YV4_WSF_STAMP=1 (bit 8 of cpr_shift): cycles per wave in [1] the counted wait for a stage, [2] fragment reads + MFMAs +
  // the stage's DMA pieces, [3] epilogue up to its vmcnt(0), [4] rest of the epilogue; printed by two workgroups
  const bool stamp_on = (cpr_shift & 256) != 0;
  // YV4_WSF_ABL (wrong results on purpose): 1 a quarter of the output stores, 2 no stage DMAs, 4 no MFMAs, 8 no epilogue,
  // 16 stage DMAs issued but out of range (zero fill, nothing fetched)
  const bool few_stores = (cpr_shift & 512) != 0;
  const bool abl_nodma = (cpr_shift & 1024) != 0, abl_nomfma = (cpr_shift & 2048) != 0, abl_noepi = (cpr_shift & 4096) != 0;
  const bool abl_oob = (cpr_shift & 8192) != 0;
  cpr_shift &= 255;
  unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tbegin = tlast;
#define YV4_WSF_STAMP(SLOT) if (stamp_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tsum[SLOT] += n_ - tlast; tlast = n_; }
#define YV4_WSF_OOB abl_oob
#define YV4_WSF_NODMA abl_nodma
#define YV4_WSF_NOMFMA abl_nomfma
#else
#define YV4_WSF_STAMP(SLOT)
#define YV4_WSF_OOB false
#define YV4_WSF_NODMA false
#define YV4_WSF_NOMFMA false
#endif
  const int kc_n = p.Cin >> 5;          // 32-channel stages per strip
  const int cpr = 1 << cpr_shift;       // 16-byte chunks per weight row (Cin / 4 >= 16)
  const int wpitch = p.Cin * 4;

  char* Ws = smem_c;
  char* ring = smem_c + BN * wpitch + wave * (kWsfStages * kWsfStageBytes);
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem;
  const unsigned ring_lds = lds_base + (unsigned)(BN * wpitch + wave * (kWsfStages * kWsfStageBytes));

  const unsigned b = blockIdx.x;
  const int xcd = (int)(b & 7u), local = (int)(b >> 3);
  const int col = local % ncol;
  const int walker = (local / ncol) * 8 + xcd;
  const int nwalkers = ((int)(gridDim.x >> 3) / ncol) * 8;
  const int NW = nwalkers * kWsfWaves;
  const int gw = walker * kWsfWaves + wave;
  const int n0 = col * BN;

  const u32x4_t rsA = make_rsrc(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc(p.w, w_bytes);

  // the weight slab, once: rows of Cin floats, 16-byte chunks XOR-swizzled with row & 15
  {
    const int groups = (BN * cpr) >> 6;
    for (int g = wave; g < groups; g += kWsfWaves) {
      const int c = g * 64 + lane;
      const int row = c >> cpr_shift;
      const int pch = c & (cpr - 1);
      const int co = n0 + row;
      const unsigned voff = co < p.Cout ? (unsigned)(((int64_t)co * p.Kw + (pch ^ (row & 15)) * 4) * 4) : kOOB;
      lds_dma16(rsB, lds_base + (unsigned)(g * 1024), voff, 0u);
    }
  }

  float s1[NT], t1[NT], s2[NT], t2[NT];
  const bool has2 = p.s2 != nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = n0 + t * 32 + r;
    const bool ok = c < p.Cout;
    s1[t] = ok ? p.s1[c] : 0.f;
    t1[t] = ok ? p.t1[c] : 0.f;
    s2[t] = (ok && has2) ? p.s2[c] : 1.f;
    t2[t] = (ok && has2) ? p.t2[c] : 0.f;
  }

  const unsigned a_rd = (unsigned)(r * 128 + ((h ^ ((r >> 1) & 7)) << 4));
  unsigned w_rd[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = t * 32 + r;
    w_rd[t] = (unsigned)(row * wpitch + ((h ^ (row & 15)) << 4));
  }

  const int my_n = gw < nstrips ? (nstrips - gw + NW - 1) / NW : 0;
  const int lrow = lane >> 3;
  const unsigned lch_even = (unsigned)(((lane & 7) ^ ((lane >> 4) & 7)) * 4);
  const unsigned lch_odd = (unsigned)(((lane & 7) ^ (((lane >> 4) + 4) & 7)) * 4);
  // Issue side: the stage DMAs of a strip start at per-lane offsets that are computed once per strip (iss_voff: rows
  // 8 j + lrow, chunk swizzled by row); the stage adds its 128 bytes through the scalar offset.  In the loop the four
  // 1 KB pieces of stage kc + 2 are issued one per j step BETWEEN the wave's own MFMAs, where a piece's issue cost
  // (~60-180 cycles, MI355X_MICROARCH.md) runs under the MFMA in flight; issued in one block at the top of the stage
  // they were 1.0-1.4 k cycles per stage that only the partner wave's MFMAs could cover (stamps: DESIGN 12.9).
  int iss_i = 0, iss_kc = 0, iss_slot = 0;
  unsigned iss_voff[4];
#define YV4_WSF_ISSUE_STRIP()                                                                               \
  {                                                                                                         \
    const int row0_ = (gw + iss_i * NW) * 32 + lrow;                                                        \
    const bool live_ = iss_i < my_n && !YV4_WSF_OOB;                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
      const int row_ = row0_ + 8 * j;                                                                       \
      const unsigned lch_ = (j & 1) ? lch_odd : lch_even;                                                   \
      iss_voff[j] = (live_ && row_ < p.M) ? (unsigned)(((int64_t)row_ * p.x_cs + p.x_co + (int)lch_) * 4) : kOOB; \
    }                                                                                                       \
  }
#define YV4_WSF_ISSUE_PIECE(J)                                                                              \
  if (!YV4_WSF_NODMA)                                                                                       \
    lds_dma16(rsA, ring_lds + (unsigned)(iss_slot * kWsfStageBytes + (J) * 1024), iss_voff[J], (unsigned)(iss_kc << 7));
#define YV4_WSF_ISSUE_ADVANCE()                                                                             \
  {                                                                                                         \
    iss_kc += 1;                                                                                            \
    if (iss_kc == kc_n) {                                                                                   \
      iss_kc = 0;                                                                                           \
      iss_i += 1;                                                                                           \
      YV4_WSF_ISSUE_STRIP();                                                                                \
    }                                                                                                       \
    iss_slot = iss_slot + 1 == kWsfStages ? 0 : iss_slot + 1;                                               \
  }
#define YV4_WSF_ISSUE()                                                                                     \
  {                                                                                                         \
    YV4_WSF_ISSUE_PIECE(0) YV4_WSF_ISSUE_PIECE(1) YV4_WSF_ISSUE_PIECE(2) YV4_WSF_ISSUE_PIECE(3)             \
    YV4_WSF_ISSUE_ADVANCE();                                                                                \
  }

  YV4_WSF_ISSUE_STRIP();
  YV4_WSF_ISSUE();
  YV4_WSF_ISSUE();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float st_su[NT], st_sq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { st_su[t] = 0.f; st_sq[t] = 0.f; }

  int rslot = 0;
  for (int i = 0; i < my_n; ++i) {
    f32x16 acc[NT], acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[t][e] = 0.f; acc2[t][e] = 0.f; }

    for (int kc = 0; kc < kc_n; ++kc) {
      YV4_WSF_STAMP(4);
      // stages 0 and 1 of a strip were confirmed in front of the previous strip's stores; later ones by count: only the
      // four pieces of stage kc + 1 (issued during stage kc - 1) may still be in flight (see conv1x1_ws_h16.hip)
      if (kc >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      YV4_WSF_STAMP(1);
      const char* st = ring + rslot * kWsfStageBytes;
      const unsigned kx = (unsigned)(kc << 7);               // (kc * 8) << 4: chunk index inside the weight row
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (YV4_WSF_NOMFMA) { YV4_WSF_ISSUE_PIECE(j) continue; }
        const float4 fa = *reinterpret_cast<const float4*>(st + (a_rd ^ (unsigned)(j << 5)));
        float4 fb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = *reinterpret_cast<const float4*>(Ws + ((w_rd[t] ^ (unsigned)(j << 5)) ^ kx));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x16& ac_ = (j & 1) ? acc2[t] : acc[t];
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[t].x, ac_, 0, 0, 0);
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[t].y, ac_, 0, 0, 0);
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[t].z, ac_, 0, 0, 0);
          ac_ = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[t].w, ac_, 0, 0, 0);
          if (t == 0) {                      // piece j of stage kc + 2, under the MFMA just issued
            __builtin_amdgcn_sched_barrier(0);
            YV4_WSF_ISSUE_PIECE(j)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      YV4_WSF_ISSUE_ADVANCE();
      rslot = rslot + 1 == kWsfStages ? 0 : rslot + 1;
      YV4_WSF_STAMP(2);
    }

    // epilogue: lane (r, h) holds channel n0 + 32t + r of pixels m0 + (e&3) + 8(e>>2) + 4h; the arithmetic is
    // epilogue_tile's, operation for operation
    const int m0 = (gw + i * NW) * 32;
    const bool full = m0 + 32 <= p.M;
#ifdef YV4_MEASURE
    if (abl_noepi) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (acc[0][0] == 12345.678f) p.y[0] = acc[NT - 1][3] + acc2[0][1];
      continue;
    }
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = n0 + t * 32 + r;
      if (c >= p.Cout) continue;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[t][e] + acc2[t][e];
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const bool in = full || (m0 + (e & 3) + 8 * (e >> 2) + 4 * h < p.M);
          st_su[t] += in ? v[e] : 0.f;
          st_sq[t] += in ? v[e] * v[e] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(v[e], s1[t], t1[t]);     // epilogue_tile's a * s + t is an fma
      act_row16(v, p.act1, p.slope1);
      if (has2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(v[e], s2[t], t2[t]);
        act_row16(v, p.act2, p.slope2);
      }
      float* yb = p.y + ((int64_t)(m0 + 4 * h) * p.y_cs + p.y_co + c);
      if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stages in flight, before the stores join the counter
      if (t == 0) { YV4_WSF_STAMP(3); }
#ifdef YV4_MEASURE
      if (few_stores) {
#pragma unroll
        for (int e = 0; e < 4; ++e) yb[(int64_t)((e & 3) + 8 * (e >> 2)) * p.y_cs] = v[e] + v[e + 4] + v[e + 8] + v[e + 12];
      } else
#endif
      if (full) {
#pragma unroll
        for (int e = 0; e < 16; ++e) yb[(int64_t)((e & 3) + 8 * (e >> 2)) * p.y_cs] = v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (m0 + (e & 3) + 8 * (e >> 2) + 4 * h < p.M) yb[(int64_t)((e & 3) + 8 * (e >> 2)) * p.y_cs] = v[e];
      }
    }
  }
#undef YV4_WSF_ISSUE
#undef YV4_WSF_ISSUE_PIECE
#undef YV4_WSF_ISSUE_ADVANCE
#undef YV4_WSF_ISSUE_STRIP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail's out-of-range stage DMAs still write this wave's ring
#ifdef YV4_MEASURE
  YV4_WSF_STAMP(4);
  if (stamp_on && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 101))
    printf("wsf wg %d wave %d strips %d: total %llu | issue %llu wait %llu reads+mfma %llu epi-to-vmcnt0 %llu epi-rest %llu (cycles)\n",
           (int)blockIdx.x, wave, my_n, __builtin_amdgcn_s_memtime() - tbegin, tsum[0], tsum[1], tsum[2], tsum[3], tsum[4]);
#endif
#undef YV4_WSF_STAMP
#undef YV4_WSF_OOB
#undef YV4_WSF_NODMA
#undef YV4_WSF_NOMFMA

  if (p.stats && my_n > 0) {
    const StatRep rep = stat_rep(p.stats, (unsigned)(gw), p.Cout);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float su = st_su[t], sq = st_sq[t];
      su += __shfl_xor(su, 32);
      sq += __shfl_xor(sq, 32);
      const int c = n0 + t * 32 + r;
      if (h == 0 && c < p.Cout) {
        stat_add(rep, c, su);
        stat_add(rep, p.Cout + c, sq);
      }
    }
  }
}

static int wsf_slab_cols(const ConvArgs& a) {
  const int cout32 = (a.Cout + 31) / 32 * 32;
  for (int bn = 128; bn >= 32; bn >>= 1) {
    if (bn > cout32) continue;
    if ((long long)bn * a.Cin * 4 + kWsfWaves * kWsfStages * kWsfStageBytes > 160 * 1024) continue;
    const int ncol = (a.Cout + bn - 1) / bn;
    if (32 % ncol != 0) continue;
    return bn;
  }
  return 0;
}

static bool conv1x1_ws_f32_applies(const ConvArgs& a) {
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.ys_on && a.res == nullptr && a.ksplit <= 1 &&
         (a.Cin == 64 || a.Cin == 128 || a.Cin == 256) && a.Kw == a.Cin && a.Cout >= 32 && wsf_slab_cols(a) > 0;
}

template <int NT>
static int launch_wsf(const ConvArgs& a, hipStream_t stream) {
  constexpr int BN = NT * 32;
  const int ncol = (a.Cout + BN - 1) / BN;
  const size_t lds = (size_t)BN * a.Cin * 4 + (size_t)kWsfWaves * kWsfStages * kWsfStageBytes;
  const int nstrips = (a.M + 31) / 32;
  int cpr_shift = 0;
  while ((4 << cpr_shift) < a.Cin) ++cpr_shift;
  const long long xb = (long long)a.N * a.H * a.W * a.x_cs * 4, wb = (long long)a.Cout * a.Kw * 4;
  auto kern = conv1x1_ws_f32_kernel<NT>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024, "conv1x1_ws_f32")) return rc;
#ifdef YV4_MEASURE
  static const int stamp = YV4_ENV_INT("YV4_WSF_STAMP", 0);
  if (stamp) cpr_shift |= 256;
  static const int abl = YV4_ENV_INT("YV4_WSF_ABL", 0);
  cpr_shift |= (abl & 31) << 9;
#endif
  hipLaunchKernelGGL(kern, dim3(kWsfGrid), dim3(kWsfThreads), lds, stream, a, (unsigned)xb, (unsigned)wb, ncol, nstrips,
                     cpr_shift);
  YV4_CHECK_LAUNCH("conv1x1_ws_f32");
  return YV4_OK;
}

static int conv1x1_ws_f32_launch(const ConvArgs& a, hipStream_t s) {
  switch (wsf_slab_cols(a)) {
    case 128: return launch_wsf<4>(a, s);
    case 64: return launch_wsf<2>(a, s);
    case 32: return launch_wsf<1>(a, s);
    default: break;
  }
  set_error("conv1x1 ws f32: no weight slab of this layer fits the LDS");
  return YV4_E_UNSUPPORTED;
}

// the pointwise layers in the kernel's domain that give each persistent wave at least YV4_WS_MINSTRIPS strips
static bool prefer_ws_f32(const ConvArgs& a) {
  static const int mode = YV4_ENV_INT("YV4_WS", 1);
  static const int min_strips = YV4_ENV_INT("YV4_WS_MINSTRIPS", 2);
  if (!mode || !conv1x1_ws_f32_applies(a)) return false;
  // strips per persistent wave: the 256 workgroups are dealt over the column slabs, 8 waves each (with the DMA pieces
  // issued under the MFMAs the kernel beats the 128x64 tile on 256 -> 256 @38 too: 62 against 67 us, tools/ab_wsf.sh)
  const int bn = wsf_slab_cols(a);
  const long long ncol = (a.Cout + bn - 1) / bn;
  return (((long long)a.M + 31) / 32) * ncol >= (long long)min_strips * 2048;
}

static bool stem_ok(const yv4_conv_desc* d, bool has_res, bool has2) {
  return d->Cin == 4 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cout <= 64 && !has_res &&
         !has2 && d->x_coff % 2 == 0 && (long long)d->N * d->H * d->W * d->x_cstride * 4 < 0xFFFFFFF0LL;
}

static int pick_tile(long long M, int Cout, bool fast_ok, long long K = 0) {
  // Static choice for callers that do not autotune (training; Plan.autotune re-decides per layer on the box it runs
  // on).  From profiles/r01_conv_shapes.txt (batch 32 YOLOv4-L shapes): deep reductions with several rounds of
  // 128x128 tiles gain 5-9 % on the big tile, the remaining K >= 576 layers and the wide 1x1 layers 3-7 % on
  // 128x64; everything else stays on 64x64 (5 workgroups per CU).
  if (fast_ok) {
    auto ntiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
    if (K >= 576 && Cout >= 128 && ntiles(128, 128) >= 700) return YV4_TILE_DMA_128x128;
    if (Cout >= 64 && (K >= 288 || Cout >= 256) && ntiles(128, 64) >= 512) return YV4_TILE_DMA_128x64;
    return YV4_TILE_DMA_64x64;
  }
  // 256 CUs x 2 resident workgroups.  Prefer the biggest tile that still gives the
  // chip >= 2 full rounds of workgroups; small maps (19x19) fall to smaller tiles.
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
  if (Cout > 64) {
    if (tiles(128, 128) >= 1024) return YV4_TILE_128x128;
    if (tiles(128, 64) >= 1024 || Cout <= 64) return YV4_TILE_128x64;
    if (tiles(64, 128) >= 768) return YV4_TILE_64x128;
    return YV4_TILE_64x64;
  }
  if (Cout > 32) return tiles(128, 64) >= 1024 ? YV4_TILE_128x64 : YV4_TILE_64x64;
  return YV4_TILE_64x64;  // Cout <= 32: half of the columns idle; stem / tiny models only
}

}  // namespace yv4

using namespace yv4;

extern "C" double yv4_conv_flops(const yv4_conv_desc* d) {
  if (!d) return 0.0;
  return 2.0 * (double)d->N * d->Ho * d->Wo * d->Cout * (double)d->KH * d->KW * d->Cin;
}

namespace yv4 {
// conv3x3_wide_f32.hip
bool conv3x3_wide_f32_applies(const ConvArgs& a);
int conv3x3_wide_f32_pick(const ConvArgs& a, double* rounds_eff);
int conv3x3_wide_f32_launch(const ConvArgs& a, int shape, hipStream_t s);

// conv_wide_f32.hip
bool conv_wide_f32_applies(const ConvArgs& a);
int conv_wide_f32_pick(const ConvArgs& a, double* rounds_eff);
int conv_wide_f32_launch(const ConvArgs& a, int shape, hipStream_t s);

// The wide-tile fp32 3x3 kernel takes the layers in its domain with >= 64 input channels and at least YV4_W3F_MINOUT
// outputs per CU when one of its tile shapes fills the rounds to within YV4_W3F_MAXWASTE percent (YV4_W3F=0: off).
static int prefer_w3_f32(const ConvArgs& a) {      // the shape index, or -1
  static const int mode = YV4_ENV_INT("YV4_W3F", 1);
  static const int waste = YV4_ENV_INT("YV4_W3F_MAXWASTE", 25);
  static const int min_out = YV4_ENV_INT("YV4_W3F_MINOUT", 32768);
  if (!mode || !conv3x3_wide_f32_applies(a) || a.Cin < 64) return -1;
  if ((long long)a.M * a.Cout < 256LL * min_out) return -1;
  double eff = 0.0;
  const int shape = conv3x3_wide_f32_pick(a, &eff);
  if (shape < 0 || eff * 100.0 > 100.0 + waste) return -1;
  return shape;
}

// The general wide-tile fp32 kernel takes the stride-2 3x3 layers with >= YV4_WGF_S2CIN input channels and the 1x1
// layers with >= YV4_WGF_P1CIN, under the same fill rule (YV4_WGF=0: off).  Measured per layer at batch 32 x 608 (round 4,
// profiles/r04_layers_wide_f32.md): stride-2 3x3 with Cin >= 128 gain 4 - 32 %, 64->128@304 and the 256-channel 1x1
// layers lose 2 - 4 % against the DMA tiles, the 512 / 1024-channel 1x1 layers gain 1 - 3 %.
static int prefer_wide_f32(const ConvArgs& a) {    // the shape index, or -1
  static const int mode = YV4_ENV_INT("YV4_WGF", 1);
  static const int waste = YV4_ENV_INT("YV4_WGF_MAXWASTE", 25);
  static const int min_out = YV4_ENV_INT("YV4_WGF_MINOUT", 32768);
  static const int s2cin = YV4_ENV_INT("YV4_WGF_S2CIN", 128);
  static const int p1cin = YV4_ENV_INT("YV4_WGF_P1CIN", 512);
  if (!mode || !conv_wide_f32_applies(a)) return -1;
  const bool s2 = a.KH == 3 && a.KW == 3 && a.stride == 2 && a.Cin >= s2cin;
  const bool p1 = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.Cin >= p1cin;
  if (!s2 && !p1) return -1;
  if ((long long)a.M * a.Cout < 256LL * min_out) return -1;
  double eff = 0.0;
  const int shape = conv_wide_f32_pick(a, &eff);
  if (shape < 0 || eff * 100.0 > 100.0 + waste) return -1;
  return shape;
}
}  // namespace yv4

extern "C" int yv4_conv_pick_tile(const yv4_conv_desc* d) {
  if (!d) return YV4_TILE_AUTO;
  const bool fast_ok = d->Cin % kBK == 0 && (long long)d->N * d->H * d->W * d->x_cstride * 4 < 0xFFFFFFF0LL &&
                       (long long)d->Cout * d->KH * d->KW * d->Cin * 4 < 0xFFFFFFF0LL;
  if (stem_ok(d, false, false)) return YV4_TILE_STEM;
  {
    ConvArgs a{};
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.Cin = d->Cin; a.Cout = d->Cout;
    a.K = a.Kw = d->KH * d->KW * d->Cin; a.M = (int)((long long)d->N * d->Ho * d->Wo);
    if (fast_ok && prefer_ws_f32(a)) return YV4_TILE_WS_1x1;     // (a residual, unknown here, keeps the tile kernels)
    a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.y_cs = d->y_cstride; a.y_co = d->y_coff;
    a.r_cs = d->r_cstride; a.r_co = d->r_coff; a.ys_on = 0;
    const int w3 = fast_ok ? prefer_w3_f32(a) : -1;
    if (w3 >= 0) return YV4_TILE_W3x3_SHAPE(w3);                  // the pinned form: a plan can copy it (see include/yv4.h)
    const int wg = fast_ok ? prefer_wide_f32(a) : -1;             // (a residual, unknown here, keeps the tile kernels)
    if (wg >= 0) return YV4_TILE_WIDE_SHAPE(wg);
  }
  return pick_tile((long long)d->N * d->Ho * d->Wo, d->Cout, fast_ok, (long long)d->KH * d->KW * d->Cin);
}

// stats != null: identity-epilogue conv that also accumulates the BatchNorm sums of its output; *stats_done tells
// the caller whether the selected kernel did (the LDS-DMA kernels do; the generic and stem kernels do not)
static int conv_f32_impl(const yv4_conv_desc* d, const float* x, const float* w, const float* scale1, const float* shift1,
                         const float* scale2, const float* shift2, const float* residual, float* y, double* stats,
                         bool* stats_done, void* stream) {
  YV4_REQUIRE(d && x && w && scale1 && shift1 && y, "conv: null argument");
  YV4_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), "conv: scale2/shift2 must come together");
  YV4_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: empty shape");
  YV4_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel/stride/pad");
  YV4_REQUIRE(d->KH * d->KW <= 64, "conv: kernels above 64 taps are not supported");
  YV4_REQUIRE(d->Cin % 4 == 0 && d->x_cstride % 4 == 0 && d->x_coff % 4 == 0,
              "conv: Cin (%d), x_cstride (%d), x_coff (%d) must be multiples of 4", d->Cin,
              d->x_cstride, d->x_coff);
  YV4_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv: x / w must be 16-byte aligned");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride, "conv: input view exceeds its pixel stride");
  YV4_REQUIRE(d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride, "conv: output view exceeds its pixel stride");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  YV4_REQUIRE(Ho == d->Ho && Wo == d->Wo, "conv: Ho/Wo (%d,%d) do not match the geometry (%d,%d)",
              d->Ho, d->Wo, Ho, Wo);
  if (residual)
    YV4_REQUIRE(d->r_coff >= 0 && d->r_coff + d->Cout <= d->r_cstride, "conv: residual view exceeds its pixel stride");
  YV4_REQUIRE(d->act1 >= 0 && d->act1 <= YV4_ACT_SWISH && d->act2 >= 0 && d->act2 <= YV4_ACT_SWISH,
              "conv: unknown activation id");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  YV4_REQUIRE(M < (1LL << 31), "conv: N*Ho*Wo = %lld does not fit 31 bits", M);
  YV4_REQUIRE((long long)d->N * d->H * d->W < (1LL << 31), "conv: N*H*W does not fit 31 bits");

  ConvArgs a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = scale2; a.t2 = shift2;
  a.res = residual; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff;
  a.r_cs = d->r_cstride; a.r_co = d->r_coff;
  a.act1 = d->act1; a.act2 = d->act2; a.slope1 = d->slope1; a.slope2 = d->slope2;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin; a.Kw = a.K; a.tiles_n = 0;
  a.ys_on = 0;
  a.stats = nullptr;
  a.ksplit = 0; a.ks_slices = 0; a.ws_cs = 0; a.ws = nullptr;

  const bool uniform = (d->Cin % kBK) == 0;
  // the LDS-DMA kernels address x and w through 32-bit buffer descriptors
  const bool fast_ok = uniform && (long long)d->N * d->H * d->W * d->x_cstride * 4 < 0xFFFFFFF0LL &&
                       (long long)d->Cout * a.K * 4 < 0xFFFFFFF0LL;
  const bool can_stem = stem_ok(d, residual != nullptr, scale2 != nullptr);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stats_done) *stats_done = false;
  {
    const int forced = (d->tile > 16 && (d->tile & 15) == YV4_TILE_W3x3 && d->tile <= YV4_TILE_W3x3_SHAPE(4)) ? (d->tile >> 4) - 1 : -1;
    if (d->tile == YV4_TILE_W3x3 || forced >= 0)
      YV4_REQUIRE(fast_ok && conv3x3_wide_f32_applies(a), "conv: the wide 3x3 tile needs a 3x3 / stride 1 / pad 1 conv with "
                  "Cin %% 32 == 0, Cout %% 16 == 0 (64 .. 1024) and 4-aligned channel strides / offsets");
    const int autoshape = (d->tile == YV4_TILE_AUTO && fast_ok) ? prefer_w3_f32(a) : -1;
    if (d->tile == YV4_TILE_W3x3 || forced >= 0 || autoshape >= 0) {
      if (stats) { a.stats = stats; *stats_done = true; }
      static const int env_shape = YV4_ENV_INT("YV4_W3F_SHAPE", -1);
      return conv3x3_wide_f32_launch(a, forced >= 0 ? forced : (autoshape >= 0 && env_shape < 0 ? autoshape : env_shape), s);
    }
  }
  {
    const int forced = (d->tile > 16 && (d->tile & 15) == YV4_TILE_WIDE && d->tile <= YV4_TILE_WIDE_SHAPE(4)) ? (d->tile >> 4) - 1 : -1;
    if (d->tile == YV4_TILE_WIDE || forced >= 0)
      YV4_REQUIRE(fast_ok && conv_wide_f32_applies(a), "conv: the wide tile needs Cin %% 32 == 0, Cout %% 16 == 0 (64 .. 2048) and "
                  "4-aligned channel strides / offsets");
    const int autoshape = (d->tile == YV4_TILE_AUTO && fast_ok && !(prefer_ws_f32(a) && conv1x1_ws_f32_applies(a))) ? prefer_wide_f32(a) : -1;
    if (d->tile == YV4_TILE_WIDE || forced >= 0 || autoshape >= 0) {
      if (stats) { a.stats = stats; *stats_done = true; }
      static const int env_shape = YV4_ENV_INT("YV4_WGF_SHAPE", -1);
      return conv_wide_f32_launch(a, forced >= 0 ? forced : (autoshape >= 0 && env_shape < 0 ? autoshape : env_shape), s);
    }
  }
  if (d->tile == YV4_TILE_WS_1x1)
    YV4_REQUIRE(fast_ok && conv1x1_ws_f32_applies(a), "conv: the weight-stationary tile needs a 1x1 / stride 1 conv with Cin 64, "
                "128 or 256, Cout >= 32 and no residual");
  if (d->tile == YV4_TILE_WS_1x1 || (d->tile == YV4_TILE_AUTO && fast_ok && prefer_ws_f32(a))) {
    if (stats) { a.stats = stats; *stats_done = true; }
    return conv1x1_ws_f32_launch(a, s);
  }
  int tile = d->tile == YV4_TILE_AUTO ? (can_stem ? YV4_TILE_STEM : pick_tile(M, d->Cout, fast_ok, a.K)) : d->tile;
  if (stats && fast_ok && (tile == YV4_TILE_DMA_64x64 || tile == YV4_TILE_DMA_128x64 || tile == YV4_TILE_DMA_128x128)) {
    a.stats = stats;
    *stats_done = true;
  }
  switch (tile) {
    case YV4_TILE_128x128: return launch_conv<128, 128, 2, 2>(a, uniform, s);
    case YV4_TILE_128x64: return launch_conv<128, 64, 2, 2>(a, uniform, s);
    case YV4_TILE_64x128: return launch_conv<64, 128, 2, 2>(a, uniform, s);
    case YV4_TILE_64x64: return launch_conv<64, 64, 2, 2>(a, uniform, s);
    case YV4_TILE_STEM: if (can_stem) return launch_conv_stem<0>(a, s); break;
    case YV4_TILE_DMA_64x64: if (fast_ok) return launch_conv_dma<64, 64, 2, 2, 2>(a, s); break;
    case YV4_TILE_DMA_128x64: if (fast_ok) return launch_conv_dma<128, 64, 2, 2, 2>(a, s); break;
    case YV4_TILE_DMA_128x128: if (fast_ok) return launch_conv_dma<128, 128, 2, 2, 2>(a, s); break;
    default:
      break;
  }
  set_error("conv: unknown or inapplicable tile id %d", tile);
  return YV4_E_INVALID;
}

extern "C" int yv4_conv_bn_act_fwd(const yv4_conv_desc* d, const float* x, const float* w,
                                   const float* scale1, const float* shift1,
                                   const float* scale2, const float* shift2,
                                   const float* residual, float* y, void* stream) {
  return conv_f32_impl(d, x, w, scale1, shift1, scale2, shift2, residual, y, nullptr, nullptr, stream);
}

// ---- split-K: the single-image (latency) form of the LDS-DMA kernels ----------------------------------------------
// At batch 1 the deep layers have a handful of tiles (512 -> 512 3x3 at 19x19: 6 x 8 tiles of 64 x 64 on 256 CUs) and
// 144 K slices each: the chip idles while 48 workgroups walk K serially.  Splitting K over `ksplit` workgroups per tile
// fills the CUs; partials go to per-split slabs (plain stores, no atomics) that one small kernel adds IN SLAB ORDER
// before the usual epilogue, so the result is deterministic (but not bit-identical to the unsplit kernel's summation
// order: plans use it for N == 1 only, where no cross-batch bit-exactness is claimed).
namespace yv4 {
__global__ __launch_bounds__(256) void splitk_finish_kernel(ConvArgs p) {
  const int c4n = p.ws_cs >> 2;
  const long long total = (long long)p.M * c4n;
  const bool has2 = p.s2 != nullptr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / c4n);
    const int co = (int)(i - (long long)m * c4n) * 4;
    const float* src = p.ws + (int64_t)m * p.ws_cs + co;
    const size_t slab = (size_t)p.M * p.ws_cs;
    float4 a = *reinterpret_cast<const float4*>(src);
    for (int s = 1; s < p.ksplit; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(src + s * slab);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = co + u;
      if (c >= p.Cout) continue;
      float t = v[u] * p.s1[c] + p.t1[c];
      t = apply_act(t, p.act1, p.slope1);
      if (p.res) t += p.res[(int64_t)m * p.r_cs + p.r_co + c];
      if (has2) t = apply_act(t * p.s2[c] + p.t2[c], p.act2, p.slope2);
      p.y[(int64_t)m * p.y_cs + p.y_co + c] = t;
    }
  }
}

// how many ways to split K for this layer on a 256-CU chip: enough workgroups to give every CU ~2, at least 4 slices
// per split; 1 = do not split (enough tiles already, or a kernel without the split path)
static int splitk_choice(const yv4_conv_desc* d, int* tile_out) {
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const int K = d->KH * d->KW * d->Cin;
  if (d->Cin % kBK != 0 || stem_ok(d, false, false)) return 1;
  if ((long long)d->N * d->H * d->W * d->x_cstride * 4 >= 0xFFFFFFF0LL || (long long)d->Cout * K * 4 >= 0xFFFFFFF0LL) return 1;
  const int tile = YV4_TILE_DMA_64x64;
  const long long tiles = ((M + 63) / 64) * ((d->Cout + 63) / 64);
  const int nk = K / kBK;
  static const long long target = (long long)YV4_ENV_INT("YV4_SPLITK_TARGET", (int)(512LL));
  static const int min_slices = YV4_ENV_INT("YV4_SPLITK_MINSL", 8);
  int ks = 1;
  while (tiles * ks < target && nk / (ks * 2) >= min_slices && ks < 32) ks *= 2;
  if (tile_out) *tile_out = tile;
  return ks;
}
}  // namespace yv4

extern "C" size_t yv4_conv_splitk_workspace(const yv4_conv_desc* d, int* ksplit) {
  if (!d) return 0;
  const int ks = splitk_choice(d, nullptr);
  if (ksplit) *ksplit = ks;
  if (ks <= 1) return 0;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  return (size_t)ks * (size_t)M * (size_t)((d->Cout + 3) / 4 * 4) * sizeof(float);
}

extern "C" int yv4_conv_bn_act_fwd_splitk(const yv4_conv_desc* d, const float* x, const float* w, const float* scale1,
                                          const float* shift1, const float* scale2, const float* shift2,
                                          const float* residual, float* y, float* workspace, size_t workspace_bytes,
                                          void* stream) {
  YV4_REQUIRE(d, "conv splitk: null descriptor");
  int tile = 0;
  const int ks = splitk_choice(d, &tile);
  if (ks <= 1) return yv4_conv_bn_act_fwd(d, x, w, scale1, shift1, scale2, shift2, residual, y, stream);
  YV4_REQUIRE(x && w && scale1 && shift1 && y && workspace, "conv splitk: null argument");
  YV4_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), "conv splitk: scale2/shift2 must come together");
  YV4_REQUIRE(d->x_cstride % 4 == 0 && d->x_coff % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 &&
              ((uintptr_t)workspace & 15) == 0, "conv splitk: alignment");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride,
              "conv splitk: view exceeds its pixel stride");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  YV4_REQUIRE(Ho == d->Ho && Wo == d->Wo, "conv splitk: Ho/Wo do not match the geometry");
  if (residual) YV4_REQUIRE(d->r_coff >= 0 && d->r_coff + d->Cout <= d->r_cstride, "conv splitk: residual view");
  YV4_REQUIRE(d->act1 >= 0 && d->act1 <= YV4_ACT_SWISH && d->act2 >= 0 && d->act2 <= YV4_ACT_SWISH, "conv splitk: activation id");
  YV4_REQUIRE(workspace_bytes >= yv4_conv_splitk_workspace(d, nullptr), "conv splitk: workspace too small");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  ConvArgs a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = scale2; a.t2 = shift2; a.res = residual; a.y = y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff;
  a.r_cs = d->r_cstride; a.r_co = d->r_coff;
  a.act1 = d->act1; a.act2 = d->act2; a.slope1 = d->slope1; a.slope2 = d->slope2;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin; a.Kw = a.K; a.tiles_n = 0; a.ys_on = 0; a.stats = nullptr;
  const int nk = a.K / kBK;
  a.ks_slices = (nk + ks - 1) / ks;
  a.ksplit = (nk + a.ks_slices - 1) / a.ks_slices;      // no empty split
  a.ws_cs = (d->Cout + 3) / 4 * 4; a.ws = workspace;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (int rc = launch_conv_dma<64, 64, 2, 2, 2>(a, s)) return rc;
  const long long work = M * (a.ws_cs / 4);
  unsigned g = (unsigned)((work + 255) / 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3(g), dim3(256), 0, s, a);
  YV4_CHECK_LAUNCH("conv splitk finish");
  return YV4_OK;
}

int conv_stats_h16(const yv4_conv_desc* d, int dtype, const void* x, const void* w, const float* ones, const float* zeros,
                   void* y, double* stats, void* stream);
int bn_partial_sums_replica0(const void* x, int dtype, int64_t M, int C, int x_cstride, int x_coff, double* stats,
                             void* stream);   // train.hip

// Training-mode convolution feeding a BatchNorm: y = conv(x, w) (identity epilogue, `ones` / `zeros` = Cout unit
// scales / zero shifts) and the per-channel sums of y for the batch statistics, accumulated by the conv kernel's
// epilogue where the selected kernel supports it, else by the BN reduction kernel afterwards -- either way
// `stats` holds YV4_STATS_REPLICAS x [sum (Cout) | sum of squares (Cout)] whose column sums are the totals.
extern "C" int yv4_conv_fwd_stats(const yv4_conv_desc* d, int dtype, const void* x, const void* w, const float* ones,
                                  const float* zeros, void* y, double* stats, int stats_is_zero, void* stream) {
  YV4_REQUIRE(d && stats, "conv_fwd_stats: null argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "conv_fwd_stats: dtype must be f32, f16 or bf16");
  YV4_REQUIRE(d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride && d->act1 == YV4_ACT_NONE,
              "conv_fwd_stats: the output view exceeds its pixel stride, or the epilogue is not the identity");
  if (!stats_is_zero &&
      hipMemsetAsync(stats, 0, sizeof(double) * YV4_STATS_REPLICAS * 2 * d->Cout, reinterpret_cast<hipStream_t>(stream)) !=
          hipSuccess) {
    set_error("conv_fwd_stats: memset failed");
    return YV4_E_LAUNCH;
  }
  // deterministic mode: the kernels receive the pointer with bit 0 set (stat_rep) and fill fixed-point replica pairs
  double* const kstats = tag_stats(stats);
  if (dtype != YV4_F32) return conv_stats_h16(d, dtype, x, w, ones, zeros, y, kstats, stream);
  bool done = false;
  const int rc = conv_f32_impl(d, reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(w), ones, zeros, nullptr,
                               nullptr, nullptr, reinterpret_cast<float*>(y), kstats, &done, stream);
  if (rc != YV4_OK || done) return rc;
  // (a tile without the statistics epilogue: one pass over y into replica 0 -- replica pair 0 in deterministic mode)
  return bn_partial_sums_replica0(y, YV4_F32, (int64_t)d->N * d->Ho * d->Wo, d->Cout, d->y_cstride, d->y_coff, stats, stream);
}

// The stem of the 16-bit path: fp32 image (NHWC, C padded to 4) and fp32 weights in, fp32 MFMA,
// output rounded to fp16 / bf16 (the layer is bound by its output bytes, which this halves).
extern "C" int yv4_conv_stem_fwd(const yv4_conv_desc* d, const float* x, const float* w, const float* scale1,
                                 const float* shift1, void* y, int out_dtype, void* stream) {
  YV4_REQUIRE(d && x && w && scale1 && shift1 && y, "conv stem: null argument");
  YV4_REQUIRE(out_dtype == YV4_F32 || out_dtype == YV4_F16 || out_dtype == YV4_BF16, "conv stem: bad out_dtype");
  YV4_REQUIRE(stem_ok(d, false, false), "conv stem: needs Cin 4 (3 padded), 3x3, stride 1, pad 1, Cout <= 64");
  YV4_REQUIRE(d->Ho == d->H && d->Wo == d->W, "conv stem: Ho/Wo must equal H/W");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride,
              "conv stem: view exceeds its pixel stride");
  YV4_REQUIRE(d->act1 >= 0 && d->act1 <= YV4_ACT_SWISH, "conv stem: unknown activation id");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  YV4_REQUIRE(M < (1LL << 31), "conv stem: N*Ho*Wo does not fit 31 bits");
  ConvArgs a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = nullptr; a.t2 = nullptr; a.res = nullptr;
  a.stats = nullptr;
  a.y = reinterpret_cast<float*>(y);
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff; a.r_cs = 0; a.r_co = 0;
  a.act1 = d->act1; a.act2 = 0; a.slope1 = d->slope1; a.slope2 = 0.f;
  a.M = (int)M; a.K = 36; a.Kw = 36; a.tiles_n = 0;
  a.ys_on = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (out_dtype == YV4_F16) return launch_conv_stem<1>(a, s);
  if (out_dtype == YV4_BF16) return launch_conv_stem<2>(a, s);
  return launch_conv_stem<0>(a, s);
}

// Convolution with a scattered store: output pixel (n, ho, wo) goes to y[n, ho*sh + oh, wo*sw + ow, :]
// of an (N, Hy, Wy, y_cstride) tensor, and d->Ho / d->Wo are taken as given (input rows past the
// bottom / right edge read zeros).  This is one parity class of the data gradient of a stride-2
// convolution: dX[2i+a, 2j+b] is a stride-1 correlation of dY with the taps of matching parity, so the
// four classes together cost exactly the forward FLOPs (a zero-dilated dY costs 4x).
extern "C" int yv4_conv_scatter_fwd(const yv4_conv_desc* d, const float* x, const float* w, const float* scale1,
                                    const float* shift1, float* y, int Hy, int Wy, int sh, int sw, int oh, int ow,
                                    void* stream) {
  YV4_REQUIRE(d && x && w && scale1 && shift1 && y, "conv scatter: null argument");
  YV4_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0, "conv scatter: empty shape");
  YV4_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 64 && d->stride == 1 && d->pad >= 0, "conv scatter: stride-1 kernels only");
  YV4_REQUIRE(d->Cin % kBK == 0 && d->x_cstride % 4 == 0 && d->x_coff % 4 == 0, "conv scatter: Cin must be a multiple of 32");
  YV4_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv scatter: x / w must be 16-byte aligned");
  YV4_REQUIRE(d->x_coff + d->Cin <= d->x_cstride && d->y_coff >= 0 && d->y_coff + d->Cout <= d->y_cstride,
              "conv scatter: view exceeds its pixel stride");
  YV4_REQUIRE(sh > 0 && sw > 0 && oh >= 0 && ow >= 0 && (d->Ho - 1) * sh + oh < Hy && (d->Wo - 1) * sw + ow < Wy,
              "conv scatter: the scattered grid does not fit the output tensor");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const long long K = (long long)d->KH * d->KW * d->Cin;
  YV4_REQUIRE(M < (1LL << 31) && (long long)d->N * d->H * d->W * d->x_cstride * 4 < 0xFFFFFFF0LL &&
              (long long)d->Cout * K * 4 < 0xFFFFFFF0LL, "conv scatter: tensors of 4 GiB or more are not supported");
  ConvArgs a;
  a.x = x; a.w = w; a.s1 = scale1; a.t1 = shift1; a.s2 = nullptr; a.t2 = nullptr; a.res = nullptr; a.y = y;
  a.stats = nullptr;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = 1; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.y_cs = d->y_cstride; a.y_co = d->y_coff; a.r_cs = 0; a.r_co = 0;
  a.act1 = 0; a.act2 = 0; a.slope1 = 0.f; a.slope2 = 0.f;
  a.M = (int)M; a.K = (int)K; a.Kw = (int)K; a.tiles_n = 0;
  a.ys_on = 1; a.ys_H = Hy; a.ys_W = Wy; a.ys_sh = sh; a.ys_sw = sw; a.ys_oh = oh; a.ys_ow = ow;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int tile = d->tile == YV4_TILE_AUTO ? YV4_TILE_DMA_64x64 : d->tile;
  switch (tile) {
    case YV4_TILE_DMA_64x64: return launch_conv_dma<64, 64, 2, 2, 2>(a, s);
    case YV4_TILE_DMA_128x64: return launch_conv_dma<128, 64, 2, 2, 2>(a, s);
    case YV4_TILE_DMA_128x128: return launch_conv_dma<128, 128, 2, 2, 2>(a, s);
    default: break;
  }
  set_error("conv scatter: tile id %d is not an LDS-DMA tile", tile);
  return YV4_E_INVALID;
}
