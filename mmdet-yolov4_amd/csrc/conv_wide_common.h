// What the four wide-tile kernels share (conv_wide_h16.hip, conv_wide_f32.hip: any kernel size / stride / padding;
// conv3x3_wide_h16.hip, conv3x3_wide_f32.hip: 3x3 / stride 1 / pad 1 with the kw-shared pixel image): the tile
// geometry, the weight image's swizzle, the epilogue (one per precision) and the whole host side -- the table of tile shapes, their LDS footprint, the
// per-layer choice (rounds of one workgroup per CU x the tile's work) and the launch.  The kernels themselves differ
// in element size, MFMA shape and in how the pixel image is fetched, and stay in their own files.
#pragma once
#include "conv_f32_common.h"
#include "conv_h16_common.h"

namespace yv4 {

constexpr int kWideThreads = 512;      // 8 waves: WAVES_M along the pixels x 8 / WAVES_M along the channels

// PT = 16-pixel tiles per wave, WAVES_M = waves along M; K3 = the 3x3 kernels' pixel image (BM + 2 pixels + one row
// that is only ever zero-filled, in whole 64-row DMA passes).  LDS rows are 128 bytes (64 16-bit / 32 fp32 channels).
template <int PT, int WAVES_M, bool K3> struct WideGeom {
  static constexpr int WAVES_N = 8 / WAVES_M;
  static constexpr int BN = 64 * WAVES_N;
  static constexpr int WMr = 16 * PT;               // pixel rows of a wave
  static constexpr int BM = WMr * WAVES_M;
  static constexpr int QA = K3 ? (BM + 3 + 63) / 64 : BM / 64;   // DMA passes (64 rows each) of the pixel image per K tile
  static constexpr int ARows = K3 ? 64 * QA : BM;
  static constexpr int ZeroRow = BM + 2;            // K3: never a source pixel
  static constexpr int PB = BN / 64;                // weight pieces per wave and K tile
  static constexpr int ABytes = ARows * 128;
  static constexpr int BBytes = BN * 128;
  static constexpr int RingBytes = 2 * ABytes + 2 * BBytes;
};

// swizzle of the weight image: the 16 lanes of a ds_read_b128 group read rows {R..R+3, R+48..R+51} at chunk q and
// {R+16..R+19, R+32..R+35} at chunk q + 1 (the kernels' channel permutation), which (row >> 1) & 7 would fold onto
// each other
// Border taps of the kw-shared 3x3 kernels: `addr` + 1 MB when bit `pos` of `nokm` is set.  A ds_read beyond the workgroup's
// LDS allocation returns zeros (tools/microbench/lds_oob_read.hip), so a lane whose tap lies outside the image needs no
// select between its row and a row of zeros -- one v_bfe_u32 per pixel tile and one v_lshl_add_u32 per read (plain C++ is
// canonicalised into shift + and + add, two instructions more per pixel tile).
template <typename T = void>
__device__ __forceinline__ unsigned wide_far_add(unsigned nokm, int pos, unsigned addr) {
  unsigned b, r;
  asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(b) : "v"(nokm), "n"(pos));
  asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(r) : "v"(b), "v"(addr));
  return r;
}

// a 16-byte fragment at LDS byte address `addr` (may lie beyond the allocation: zeros)
template <typename V>
__device__ __forceinline__ V wide_lds_read(unsigned addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) V*>(static_cast<uintptr_t>(addr));
}

// the same with the bit position in a register or scalar (the fp32 kernel's tap bit 3 kh + kw)
template <typename T = void>
__device__ __forceinline__ unsigned wide_far_bit(unsigned nokm, int pos) {
  unsigned b;
  asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(b) : "v"(nokm), "s"(pos));
  return b;
}
template <typename T = void>
__device__ __forceinline__ unsigned wide_far_add_bit(unsigned b, unsigned addr) {
  unsigned r;
  asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(r) : "v"(b), "v"(addr));
  return r;
}

__device__ __forceinline__ int wide_swz_b(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

// ---- epilogue, shared by the general and the 3x3 kernel of a precision -------------------------------------------------
// A lane owns ONE pixel of each of the wave's PT 16-pixel tiles (`mlane` + 16 pt) and 16 consecutive channels
// (`cl` .. `cl` + 15, accumulators acc[pt][t][j] = channel cl + 4 t + j): the epilogue reads and writes whole 32 / 64-byte
// segments, no LDS transposition and no lane exchange.  `aff` = the layer's [scale1 | shift1 | scale2 | shift2] in LDS.
// Expressions and their order are the generic tiles' (conv_mfma_f32.hip's epilogue_tile, conv_h16_common.h's):
// fmaf(acc, s1, t1) -> act -> + residual -> fmaf(., s2, t2) -> act.  SCATTER: the output row of pixel m goes through
// out_row (the general kernels' upsample-scatter view); `replica` = the wave's row slab for the BatchNorm statistics
// (yv4_conv_fwd_stats).
typedef float wide_acc_t __attribute__((ext_vector_type(4)));

// BatchNorm statistics of the tile: st = [sums | sums of squares] of the lane's 16 channels over its pixels.  The 16
// lanes that share a channel group hold partial sums of the same 32 quantities over different pixels -- a halving
// butterfly over lane bits 0..3 (16 + 8 + 4 + 2 exchanges) leaves two finished sums per lane.  (A macro: as a function
// taking the array by reference it keeps `st` in scratch memory.)
#define YV4_WIDE_STATS_FLUSH(st, lane, c_ok, stats, replica, Cout, cl)                                   \
  {                                                                                                      \
    int idx_ = 0;                                                                                        \
    _Pragma("unroll") for (int sft = 0; sft < 4; ++sft) {                                                \
      const int half = 16 >> sft;                                                                        \
      const bool bit = ((lane) >> sft) & 1;                                                              \
      _Pragma("unroll") for (int i = 0; i < half; ++i) {                                                 \
        const float send = bit ? st[i] : st[i + half];                                                   \
        const float recv = __shfl_xor(send, 1 << sft);                                                   \
        st[i] = (bit ? st[i + half] : st[i]) + recv;                                                     \
      }                                                                                                  \
      idx_ += bit ? half : 0;                                                                            \
    }                                                                                                    \
    if (c_ok) {                                                                                          \
      const StatRep rep = stat_rep((stats), (replica), (Cout));                                          \
      _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                    \
        const int id = idx_ + k; /* 0..15: sums, 16..31: sums of squares, of channel cl + (id & 15) */   \
        stat_add(rep, (id >> 4) * (Cout) + (cl) + (id & 15), st[k]);                                     \
      }                                                                                                  \
    }                                                                                                    \
  }

__device__ __forceinline__ void wide_load_affine1(const float* aff, int Cout, int ca, float (&s1)[16], float (&t1)[16]) {
#pragma unroll
  for (int u = 0; u < 16; u += 4) {
    const float4 a = *reinterpret_cast<const float4*>(aff + ca + u), b = *reinterpret_cast<const float4*>(aff + Cout + ca + u);
    s1[u] = a.x; s1[u + 1] = a.y; s1[u + 2] = a.z; s1[u + 3] = a.w;
    t1[u] = b.x; t1[u + 1] = b.y; t1[u + 2] = b.z; t1[u + 3] = b.w;
  }
}

// 16-bit: packed activation rows (act_row8), two 16-byte stores per pixel; the statistics are those of the STORED values
// RD = pixel tiles whose residual is requested ahead of their turn.  Requesting ALL of them before the first value is
// finished was built on the hypothesis that a cold epilogue is PT memory round trips in a row; the stamps say it is not
// (profiles/r06_w3_stamps_cold.txt: the same 17-40 k cycles either way -- it is the CU's ~24 GB/s to HBM), and the registers
// it holds cost the network 0.7 % (profiles/r06_ab_wide_rd.txt: bf16 inference 4 780-4 800 with RD = PT, 4 814-4 822 with 1).
#ifdef YV4_WIDE_RD_ALL         /* A/B build: RD = all tiles (four at eight pixel tiles) */
#define YV4_WIDE_RD_DEFAULT(PT) ((PT) <= 6 ? (PT) : 4)
#else
#define YV4_WIDE_RD_DEFAULT(PT) 1
#endif
template <bool BF16, int PT, bool SCATTER, int RD = YV4_WIDE_RD_DEFAULT(PT)>
__device__ __forceinline__ void wide_epilogue_h16(const ConvArgsH& p, const float* aff, bool has2, const wide_acc_t (&acc)[PT][4],
                                                  int mlane, int cl, int lane, unsigned replica) {
  typedef typename Elem<BF16>::V8 V8;
  typedef typename Elem<BF16>::T T;
  const bool c_ok = cl + 15 < p.Cout;
  const int ca = c_ok ? cl : 0;
  float s1[16], t1[16];
  wide_load_affine1(aff, p.Cout, ca, s1, t1);
  // The residual of the next RD pixel tiles is requested ahead (RD = 1: tile by tile; see YV4_WIDE_RD_DEFAULT).
  constexpr int D = RD;
  V8 rres[D][2] = {};
#define YV4_WIDE_RES_LOAD(pt_)                                                                        \
  {                                                                                                  \
    const int m_ = mlane + 16 * (pt_);                                                               \
    if (c_ok && m_ < p.M) {                                                                          \
      const T* rp = reinterpret_cast<const T*>(p.res) + (int64_t)m_ * p.r_cs + p.r_co + cl;          \
      rres[(pt_) % D][0] = *reinterpret_cast<const V8*>(rp);                                         \
      rres[(pt_) % D][1] = *reinterpret_cast<const V8*>(rp + 8);                                     \
    }                                                                                                \
  }
  if (p.res) {
#pragma unroll
    for (int pt = 0; pt < D; ++pt) YV4_WIDE_RES_LOAD(pt)
    __builtin_amdgcn_sched_barrier(0);
  }
  float st[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) st[u] = 0.f;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int m = mlane + 16 * pt;
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
    if (p.res) {
      if (ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] += (float)rres[pt % D][0][e]; v[e + 8] += (float)rres[pt % D][1][e]; }
      }
      if (pt + D < PT) {                       // the slot is free: the tile D ahead
        __builtin_amdgcn_sched_barrier(0);
        YV4_WIDE_RES_LOAD(pt + D)
        __builtin_amdgcn_sched_barrier(0);
      }
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
      T* yp = reinterpret_cast<T*>(p.y) + (SCATTER ? out_row_h(p, m) : (int64_t)m) * p.y_cs + p.y_co + cl;
      if (p.nt_out) {                       // (a per-plan choice, include/yv4.h YV4_CONV_NT_OUT; wave-uniform)
        __builtin_nontemporal_store(o0, reinterpret_cast<V8*>(yp));
        __builtin_nontemporal_store(o1, reinterpret_cast<V8*>(yp + 8));
      } else {
        *reinterpret_cast<V8*>(yp) = o0;
        *reinterpret_cast<V8*>(yp + 8) = o1;
      }
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
  if (p.stats) YV4_WIDE_STATS_FLUSH(st, lane, c_ok, p.stats, replica, p.Cout, cl)
#undef YV4_WIDE_RES_LOAD
}

// fp32: the contraction-free scalar activations of the fp32 kernels (apply_act), four 16-byte stores per pixel
template <int PT, bool SCATTER>
__device__ __forceinline__ void wide_epilogue_f32(const ConvArgs& p, const float* aff, bool has2, const wide_acc_t (&acc)[PT][4],
                                                  int mlane, int cl, int lane, unsigned replica) {
  const bool c_ok = cl + 15 < p.Cout;
  const int ca = c_ok ? cl : 0;
  float s1[16], t1[16];
  wide_load_affine1(aff, p.Cout, ca, s1, t1);
  float st[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) st[u] = 0.f;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int m = mlane + 16 * pt;
    const bool ok = c_ok && m < p.M;
    float v[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * t + j] = apply_act(__builtin_fmaf(acc[pt][t][j], s1[4 * t + j], t1[4 * t + j]), p.act1, p.slope1);
    if (p.res && ok) {
      const float* rp = p.res + (int64_t)m * p.r_cs + p.r_co + cl;
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        const float4 r4 = *reinterpret_cast<const float4*>(rp + u);
        v[u] += r4.x; v[u + 1] += r4.y; v[u + 2] += r4.z; v[u + 3] += r4.w;
      }
    }
    if (has2) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = apply_act(__builtin_fmaf(v[u], aff[2 * p.Cout + ca + u], aff[3 * p.Cout + ca + u]), p.act2, p.slope2);
    }
    if (ok) {
      float* yp = p.y + (SCATTER ? out_row(p, m) : (int64_t)m) * p.y_cs + p.y_co + cl;
#pragma unroll
      for (int u = 0; u < 16; u += 4) *reinterpret_cast<float4*>(yp + u) = make_float4(v[u], v[u + 1], v[u + 2], v[u + 3]);
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { st[e] += v[e]; st[16 + e] += v[e] * v[e]; }
      }
    }
  }
  if (p.stats) YV4_WIDE_STATS_FLUSH(st, lane, c_ok, p.stats, replica, p.Cout, cl)
}

// ---- host side --------------------------------------------------------------------------------------------------------
// Tile shapes (pixel tiles per wave, waves along M): (8,2) 256 x 256, (6,2) 192 x 256, (4,2) 128 x 256, (6,4) 384 x 128,
// (4,4) 256 x 128; the 16-bit 3x3 kernel also (3,8) 384 x 64 and (2,8) 256 x 64 for layers with 64 output channels.  The ids
// are part of the boundary (YV4_TILE_* shape arguments, include/yv4.h).
struct WideShape { int pt, wm; };
constexpr int kWideShapesGeneral = 5, kWideShapes3x3H = 7;
static const WideShape kWideShapes[7] = {{8, 2}, {6, 2}, {4, 2}, {6, 4}, {4, 4}, {3, 8}, {2, 8}};

inline int wide_cus() {
  static std::atomic<int> g{0};
  int cus = g.load(std::memory_order_relaxed);
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
    g.store(cus, std::memory_order_relaxed);
  }
  return cus;
}

// ring (two pixel images + two weight images of 128-byte rows) + the layer's two affines (4 floats per channel)
inline size_t wide_lds(bool k3, int pt, int wmv, int Cout) {
  const int bm = 16 * pt * wmv, bn = 64 * (8 / wmv);
  const int arows = k3 ? 64 * ((bm + 3 + 63) / 64) : bm;
  return (size_t)2 * arows * 128 + (size_t)2 * bn * 128 + (size_t)16 * Cout;
}

// Cost model = rounds of one workgroup per CU x the tile's work.  The 16-bit kernels charge the smaller wave tiles for
// their extra fragment reads per MFMA (measured ratios, tools/conv_bench.py: 1.04 at six pixel tiles, 1.12 at four);
// the fp32 kernels are matrix-bound and do not (ties go to the first shape).  Returns an index into kWideShapes, or -1
// when no shape fits the LDS; *rounds_eff = cost relative to a perfectly divided layer.
// The 16-bit 3x3 kernel (round 6: LOAD / MFMA intervals between SIMD partners) is chosen by a TIME model instead, fitted
// on profiles/r06_w3_shapes.txt (batch 32, bf16): a round of tiles costs F + bm bn K c with (F us, c us per 1e6) =
// (14, 0.311) at eight pixel tiles per wave, (10, 0.353) at six, (4.5, 0.417) at four -- the big wave tile has the
// cheapest K loop and the most expensive epilogue.  *rounds_eff keeps its meaning (tile work incl. the old read weights).
template <class Args>
inline int wide_pick(const Args& a, bool k3, bool charge_reads, double* rounds_eff, int nshapes = kWideShapesGeneral) {
  const int cus = wide_cus();
#ifdef YV4_WIDE_PICK_OLD        /* A/B build: the work model of rounds 4-5 for every kernel */
  const bool time_model = false;
#else
  const bool time_model = k3 && charge_reads;
#endif
  int best = -1;
  double best_cost = 0.0, best_eff = 0.0;
  for (int i = 0; i < nshapes; ++i) {
    const int pt = kWideShapes[i].pt, wmv = kWideShapes[i].wm;
    const int bm = 16 * pt * wmv, bn = 64 * (8 / wmv);
    if (wide_lds(k3, pt, wmv, a.Cout) > 160 * 1024) continue;
    if (bn > ((a.Cout + 127) / 128) * 128) continue;                  // a 256-wide tile on a 128-channel layer is half empty
    const long long tiles = ((long long)a.M + bm - 1) / bm * ((a.Cout + bn - 1) / bn);
    const long long rounds = (tiles + cus - 1) / cus;
    const double eff = !charge_reads || pt == 8 ? 1.0 : (pt == 6 ? 1.04 : (pt == 4 ? 1.12 : 1.2));
    const double work = (double)rounds * bm * bn * eff;
    double cost = work;
    if (time_model) {
      const double F = pt == 8 ? 14.0 : (pt == 6 ? 10.0 : 4.5), c = pt == 8 ? 0.311 : (pt == 6 ? 0.353 : (pt == 4 ? 0.417 : 0.46));
      cost = (double)rounds * (F + (double)bm * bn * 9.0 * a.Cin * c * 1e-6);
    }
    if (best < 0 || cost < best_cost * (charge_reads ? 1.0 : 0.999)) { best = i; best_cost = cost; best_eff = work; }
  }
  if (rounds_eff && best >= 0) *rounds_eff = best_eff / ((double)a.M * a.Cout / cus);
  return best;
}

inline bool wide_shape_fits(const char* who, bool k3, int shape, int Cout, int nshapes = kWideShapesGeneral) {
  if (shape < 0 || shape >= nshapes || wide_lds(k3, kWideShapes[shape].pt, kWideShapes[shape].wm, Cout) > 160 * 1024) {
    set_error("%s: no tile shape of this layer fits the LDS", who);
    return false;
  }
  return true;
}

// One persistent workgroup per CU (fewer when the layer has fewer tiles) walks the tiles.  `esize` = bytes per
// element of x and w: both are addressed through 32-bit buffer descriptors.
template <class G_, class Args, class Kern>
inline int wide_launch(Kern kern, LdsAttrOnce& once, const char* who, const Args& a, int esize, hipStream_t stream) {
  Args p = a;
  const int tiles_m = (p.M + G_::BM - 1) / G_::BM;
  p.tiles_n = (p.Cout + G_::BN - 1) / G_::BN;
  p.fd_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
  p.fd_wo = make_fastdiv((unsigned)p.Wo);
  const long long tiles = (long long)tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("%s: grid of %lld tiles out of range", who, tiles);
    return YV4_E_INVALID;
  }
  const size_t lds = (size_t)G_::RingBytes + (size_t)4 * p.Cout * 4;
  if (lds > 160 * 1024) {
    set_error("%s: %zu bytes of LDS for this tile shape and Cout", who, lds);
    return YV4_E_UNSUPPORTED;
  }
  const long long xb = (long long)p.N * p.H * p.W * p.x_cs * esize, wb = (long long)p.Cout * p.Kw * esize;
  if (xb >= 0xFFFFFFF0LL || wb >= 0xFFFFFFF0LL) {
    set_error("%s: tensors of 4 GiB or more are not addressable through a buffer descriptor", who);
    return YV4_E_UNSUPPORTED;
  }
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024, who)) return rc;
  const int cus = wide_cus();
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kWideThreads), lds, stream, p, (unsigned)xb, (unsigned)wb, (int)tiles);
  YV4_CHECK_LAUNCH(who);
  return YV4_OK;
}

}  // namespace yv4
