// Shared pieces of the 16-bit fused convolution kernels (conv_mfma_h16.hip: the generic implicit-GEMM tiles;
// conv3x3_h16.hip: the 3x3 / stride-1 kernel that keeps its im2col rows in LDS): argument block, element traits,
// LDS-DMA helpers and the fused epilogue of one 32x32 accumulator tile.
#pragma once
#include "yv4_common.h"

namespace yv4 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kHBK = 64;        // K slice in elements (128 bytes)
constexpr int kHThreads = 256;

struct ConvArgsH {
  const void* x;
  const void* w;
  const float* s1;
  const float* t1;
  const float* s2;
  const float* t2;
  const void* res;
  void* y;
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int x_cs, x_co, y_cs, y_co, r_cs, r_co;
  int act1, act2;
  float slope1, slope2;
  int M, K, Kw;
  int tiles_n;
  int out_f32;    // store fp32 instead of the operand type (pred maps feeding the fp32 decode kernel)
  int ys_on, ys_H, ys_W, ys_sh, ys_sw, ys_oh, ys_ow;   // scattered output, see conv_mfma_f32.hip
  int nt_out;     // yv4_conv_desc.flags & YV4_CONV_NT_OUT: non-temporal output stores (the wide-tile epilogue)
  int ablate;     // measurement only (YV4_H16_ABLATE): 1 = issue no DMA after the first slice, 2 = no MFMA, 4 = no barrier
  double* stats;  // training: per-channel [sum | sum of squares] of the STORED outputs, YV4_STATS_REPLICAS x 2*Cout
  FastDiv fd_hw, fd_wo;   // m / (Ho*Wo), r / Wo (set by launch_h16)
  FastDiv fd_cin, fd_kw;  // GENERAL_K: k / Cin, tap / KW, once per lane per slice; fd_kw also: tap -> (kh, kw) of a split
  // split-K (single-image plans, uniform-K tiles only): workgroup (tile, split) reduces K slices [split * ks_slices, ...)
  // and stores its RAW fp32 partial tile into slab `split` of ws ([ksplit][M][ws_cs]); splitk_finish_h16_kernel adds the
  // slabs in slab order and applies the epilogue.  ksplit <= 1: off.
  int ksplit, ks_slices, ws_cs;
  float* ws;
  FastDiv fd_taps;        // slice -> (chunk, tap) at a split's first slice
};

__device__ __forceinline__ int64_t out_row_h(const ConvArgsH& p, int m) {
  if (!p.ys_on) return m;
  const int hw = p.Ho * p.Wo;
  const int n = m / hw;
  const int r = m - n * hw;
  const int ho = r / p.Wo;
  const int wo = r - ho * p.Wo;
  return ((int64_t)n * p.ys_H + ho * p.ys_sh + p.ys_oh) * p.ys_W + wo * p.ys_sw + p.ys_ow;
}

__device__ __forceinline__ void lds_dma16_h(u32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ u32x4_t make_rsrc_h(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  u32x4_t v;
  v.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  v.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  v.z = __builtin_amdgcn_readfirstlane(bytes);
  v.w = 0x00020000u;
  return v;
}

template <bool BF16>
struct Elem;
template <>
struct Elem<true> {
  typedef __bf16 T;
  typedef bf16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Elem<false> {
  typedef _Float16 T;
  typedef f16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// Epilogue of one 32x32 accumulator tile through a wave-private LDS patch (fp32, pitch 36):
// afterwards lane l owns 8 consecutive channels of rows (l>>2) and (l>>2)+16, i.e. one 16-byte
// store of 16-bit outputs per row (two dwordx4 when the output is fp32).
// the activation of 8 values with ONE (uniform) branch on the activation id: apply_act() inside the element
// loop left a scalar compare-and-branch chain per element in the epilogue (the compiler does not unswitch it)
__device__ __forceinline__ void act_row8(float (&v)[8], int act, float slope) {
  switch (act) {
    case YV4_ACT_MISH:
      mish_fast_row(v);       // packed pairs (yv4_common.h)
      break;
    case YV4_ACT_LEAKY:
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = v[u] >= 0.f ? v[u] : v[u] * slope;
      break;
    case YV4_ACT_SWISH:
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = apply_act(v[u], YV4_ACT_SWISH, 0.f);
      break;
    default:
      break;
  }
}

// Stores from the MFMA's C layout (lane (r, h) = output channel r of the rows (e & 3) + 8 (e >> 2) + 4 h): a 16-bit
// value per lane and store is a sub-dword write per lane, which the memory pipeline handles at a fraction of the dword
// rate (measured: the persistent kernels that store this way were bound by it).  Two lanes that own adjacent
// channels (r even / odd) of the same 16 rows trade halves through one DPP swap per value pair, so that each stores
// full dwords: afterwards out[j] (j = 0..7) is, on the even lane, channels (c, c+1) of row (j & 3) + 8 (j >> 2) + 4 h,
// on the odd lane channels (c-1, c) of row 16 + (j & 3) + 8 (j >> 2) + 4 h.
template <typename T>
__device__ __forceinline__ void pair_pack16(const float (&v)[16], bool odd, unsigned (&out)[8]) {
  typedef T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float send = odd ? v[j] : v[j + 8];
    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
    T2 pk;
    pk[0] = (T)(odd ? recv : v[j]);
    pk[1] = (T)(odd ? v[j + 8] : recv);
    out[j] = __builtin_bit_cast(unsigned, pk);
  }
}

// per-channel affine of a lane's 8 output columns (co .. co+7): two 16-byte loads per array when aligned
struct AffH { float s1[8], t1[8], s2[8], t2[8]; };
__device__ __forceinline__ void load_affine_h(const ConvArgsH& p, int co, bool has2, AffH& a) {
  const bool al = (((uintptr_t)p.s1 | (uintptr_t)p.t1 | (uintptr_t)p.s2 | (uintptr_t)p.t2) & 15) == 0 && (co & 3) == 0;
  if (al) {
    const float4 a0 = *reinterpret_cast<const float4*>(p.s1 + co), a1 = *reinterpret_cast<const float4*>(p.s1 + co + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.t1 + co), b1 = *reinterpret_cast<const float4*>(p.t1 + co + 4);
    a.s1[0] = a0.x; a.s1[1] = a0.y; a.s1[2] = a0.z; a.s1[3] = a0.w; a.s1[4] = a1.x; a.s1[5] = a1.y; a.s1[6] = a1.z; a.s1[7] = a1.w;
    a.t1[0] = b0.x; a.t1[1] = b0.y; a.t1[2] = b0.z; a.t1[3] = b0.w; a.t1[4] = b1.x; a.t1[5] = b1.y; a.t1[6] = b1.z; a.t1[7] = b1.w;
    if (has2) {
      const float4 c0 = *reinterpret_cast<const float4*>(p.s2 + co), c1 = *reinterpret_cast<const float4*>(p.s2 + co + 4);
      const float4 d0 = *reinterpret_cast<const float4*>(p.t2 + co), d1 = *reinterpret_cast<const float4*>(p.t2 + co + 4);
      a.s2[0] = c0.x; a.s2[1] = c0.y; a.s2[2] = c0.z; a.s2[3] = c0.w; a.s2[4] = c1.x; a.s2[5] = c1.y; a.s2[6] = c1.z; a.s2[7] = c1.w;
      a.t2[0] = d0.x; a.t2[1] = d0.y; a.t2[2] = d0.z; a.t2[3] = d0.w; a.t2[4] = d1.x; a.t2[5] = d1.y; a.t2[6] = d1.z; a.t2[7] = d1.w;
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.s2[u] = 1.f; a.t2[u] = 0.f; }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a.s1[u] = p.s1[co + u];
      a.t1[u] = p.t1[co + u];
      a.s2[u] = has2 ? p.s2[co + u] : 1.f;
      a.t2[u] = has2 ? p.t2[co + u] : 0.f;
    }
  }
}

// `full` (lane-uniform per tile column group): the vector path applies, `af` holds the lane's affine
// STAGE / FINISH split the function for callers that give every accumulator tile of a wave its own patch: all tiles
// are staged first (their LDS writes pipeline), then finished -- the LDS round trip is paid once per wave instead of
// once per tile (conv3x3_h16.hip, where nothing else on the CU hides it).
template <bool BF16, bool STAGE = true, bool FINISH = true>
__device__ __forceinline__ void epilogue_tile_h(const ConvArgsH& p, const f32x16& acc, float* ep, int lane, int m_base,
                                                int co_base, bool full, bool has2, const AffH& af) {
  typedef typename Elem<BF16>::T T;
  typedef typename Elem<BF16>::V8 V8;
  const int r = lane & 31, h = lane >> 5;
  constexpr int kPitch = 36;
  if (STAGE) {
#pragma unroll
    for (int e = 0; e < 16; ++e) ep[((e & 3) + 8 * (e >> 2) + 4 * h) * kPitch + r] = acc[e];
  }
  if (!FINISH) return;
  const int c8 = (lane & 3) * 8;
  const int co = co_base + c8;
  if (p.ksplit > 1) {                       // split-K partial: raw accumulators into a dense [M][ws_cs] slab (p.y = the slab)
    if (co < p.ws_cs) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = (lane >> 2) + 16 * k;
        const int m = m_base + row;
        if (m < p.M) {
          float* dst = reinterpret_cast<float*>(p.y) + (int64_t)m * p.ws_cs + co;
          *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(ep + row * kPitch + c8);
          *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(ep + row * kPitch + c8 + 4);
        }
      }
    }
    return;
  }
  if (full) {
    const float (&s1)[8] = af.s1; const float (&t1)[8] = af.t1; const float (&s2)[8] = af.s2; const float (&t2)[8] = af.t2;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = (lane >> 2) + 16 * k;
      const int m = m_base + row;
      const float4 a0 = *reinterpret_cast<const float4*>(ep + row * kPitch + c8);
      const float4 a1 = *reinterpret_cast<const float4*>(ep + row * kPitch + c8 + 4);
      if (m < p.M) {
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_fmaf(v[u], s1[u], t1[u]);     // (explicit: every 16-bit conv kernel rounds the affine the same way)
        act_row8(v, p.act1, p.slope1);
        if (p.res) {
          const V8 rr = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.res) + (int64_t)m * p.r_cs + p.r_co + co);
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] += (float)rr[u];
        }
        if (has2) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = __builtin_fmaf(v[u], s2[u], t2[u]);
          act_row8(v, p.act2, p.slope2);
        }
        if (YV4_ABLATE(p.ablate, 16)) {
          if (v[0] == 12345.678f) reinterpret_cast<float*>(p.y)[0] = v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
        } else if (p.out_f32) {
          float* dst = reinterpret_cast<float*>(p.y) + out_row_h(p, m) * p.y_cs + p.y_co + co;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          V8 o;
#pragma unroll
          for (int u = 0; u < 8; ++u) o[u] = (T)v[u];
          *reinterpret_cast<V8*>(reinterpret_cast<T*>(p.y) + out_row_h(p, m) * p.y_cs + p.y_co + co) = o;
        }
      }
    }
  } else if (p.out_f32) {
    // fp32 rows that are not 16-byte aligned (the 255-channel pred maps): lane (r, h) takes channel co_base + r of the
    // rows 2k + h, so a store instruction writes two runs of 32 consecutive floats -- coalesced whatever the row's
    // alignment (eight scattered dwords per lane and instruction took 3.7x the pred convs' byte floor)
    const int c = co_base + r;
    if (c < p.Cout) {
      const float sc1 = p.s1[c], sh1 = p.t1[c];
      const float sc2 = has2 ? p.s2[c] : 1.f, sh2 = has2 ? p.t2[c] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = 2 * k + h;
        const int m = m_base + row;
        if (m >= p.M) continue;
        float v = __builtin_fmaf(ep[row * kPitch + r], sc1, sh1);
        v = apply_act(v, p.act1, p.slope1);
        if (p.res) v += (float)reinterpret_cast<const T*>(p.res)[(int64_t)m * p.r_cs + p.r_co + c];
        if (has2) v = apply_act(__builtin_fmaf(v, sc2, sh2), p.act2, p.slope2);
        reinterpret_cast<float*>(p.y)[out_row_h(p, m) * p.y_cs + p.y_co + c] = v;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = (lane >> 2) + 16 * k;
      const int m = m_base + row;
      if (m >= p.M) continue;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = co + u;
        if (c < p.Cout) {
          float v = __builtin_fmaf(ep[row * kPitch + c8 + u], p.s1[c], p.t1[c]);
          v = apply_act(v, p.act1, p.slope1);
          if (p.res) v += (float)reinterpret_cast<const T*>(p.res)[(int64_t)m * p.r_cs + p.r_co + c];
          if (has2) v = apply_act(__builtin_fmaf(v, p.s2[c], p.t2[c]), p.act2, p.slope2);
          if (p.out_f32)
            reinterpret_cast<float*>(p.y)[out_row_h(p, m) * p.y_cs + p.y_co + c] = v;
          else
            reinterpret_cast<T*>(p.y)[out_row_h(p, m) * p.y_cs + p.y_co + c] = (T)v;
        }
      }
    }
  }
}

// ---- LDS-free epilogue of the persistent 3x3 kernels (conv3x3_pp_h16.hip, conv3x3_sw_h16.hip) --------------------------
// Epilogue of one 32x32 accumulator tile straight from the MFMA's C layout (lane (r, h): channel co_base + r of the
// rows (e & 3) + 8 (e >> 2) + 4 h).  Expressions and their order are epilogue_tile_h's (conv_h16_common.h).  `aff` is
// the layer's affine in LDS ([s1 | t1 | s2 | t2] x Cout); `resw` the residual words of this tile, requested by the caller
// for all of the wave's tiles before the first is finished (one memory round trip per tile of the grid, not per value).
template <bool BF16>
__device__ __forceinline__ void residual_prefetch_h(const ConvArgsH& p, int lane, int m_base, int co_base, unsigned (&resw)[8]) {
  typedef typename Elem<BF16>::T T;
  const int r = lane & 31, h = lane >> 5;
  const bool odd = r & 1;
  const int cp = co_base + r - (odd ? 1 : 0);
  const bool c_ok = cp + 1 < p.Cout;
  const int row0 = m_base + 4 * h + (odd ? 16 : 0);
  const T* rp = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = row0 + (j & 3) + 8 * (j >> 2);
    resw[j] = (c_ok && m < p.M) ? *reinterpret_cast<const unsigned*>(rp + (int64_t)m * p.r_cs + p.r_co + cp) : 0u;
  }
}

template <bool BF16>
__device__ __forceinline__ void epilogue_pairs_h(const ConvArgsH& p, const f32x16& acc, int lane, int m_base, int co_base,
                                                 bool has2, const float* aff, const unsigned (&resw)[8]) {
  typedef typename Elem<BF16>::T T;
  typedef T T2 __attribute__((ext_vector_type(2)));
  const int r = lane & 31, h = lane >> 5;
  const bool odd = r & 1;
  const int c = co_base + r;
  const int cp = c - (odd ? 1 : 0);                  // even channel of this lane's pair
  const bool c_ok = cp + 1 < p.Cout;                 // Cout is even in this kernel's domain
  const int cc = c_ok ? c : 0;
  const float s1 = aff[cc], t1 = aff[p.Cout + cc];
  // rows this lane stores after the exchange: (j & 3) + 8 (j >> 2) + 4 h (+ 16 on odd lanes)
  const int row0 = m_base + 4 * h + (odd ? 16 : 0);
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(acc[e], s1, t1);
  {
    float lo[8], hi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { lo[e] = v[e]; hi[e] = v[e + 8]; }
    act_row8(lo, p.act1, p.slope1);
    act_row8(hi, p.act1, p.slope1);
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = lo[e]; v[e + 8] = hi[e]; }
  }
  // exchange: the even lane keeps its rows e < 8 and receives the odd lane's, the odd lane keeps e >= 8
  float a[8], b[8];                                  // channel cp, channel cp + 1 of row j
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float send = odd ? v[j] : v[j + 8];
    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
    a[j] = odd ? recv : v[j];
    b[j] = odd ? v[j + 8] : recv;
  }
  if (p.res) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const T2 rr = __builtin_bit_cast(T2, resw[j]);
      a[j] += (float)rr[0];
      b[j] += (float)rr[1];
    }
  }
  if (has2) {
    const int c2 = c_ok ? cp : 0;
    const float s2a = aff[2 * p.Cout + c2], t2a = aff[3 * p.Cout + c2], s2b = aff[2 * p.Cout + c2 + 1], t2b = aff[3 * p.Cout + c2 + 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = __builtin_fmaf(a[j], s2a, t2a);
      b[j] = __builtin_fmaf(b[j], s2b, t2b);
    }
    act_row8(a, p.act2, p.slope2);
    act_row8(b, p.act2, p.slope2);
  }
  T* yp = reinterpret_cast<T*>(p.y);
  T* y0 = yp + (int64_t)row0 * p.y_cs + p.y_co + cp;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int dj = (j & 3) + 8 * (j >> 2);
    T2 pk;
    pk[0] = (T)a[j];
    pk[1] = (T)b[j];
    if (c_ok && row0 + dj < p.M) *reinterpret_cast<unsigned*>(y0 + (int64_t)dj * p.y_cs) = __builtin_bit_cast(unsigned, pk);
  }
}

}  // namespace yv4
