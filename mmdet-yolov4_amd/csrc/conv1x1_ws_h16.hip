// 1x1 / stride-1 fused convolution for gfx950 with 16-bit operands, weight-stationary and persistent -- the HBM-bound
// pointwise layers of CSPDarknet53 / PAN (darknetcsp.py:38-64 bottleneck cv1, :192-228 CSP cv1..cv4; their data
// gradients are 1x1 convolutions too).
//
// Why a third kernel.  A 1x1 layer with Cin <= 256 is a GEMM with a K of one to four 64-element slices: 2-10 us of
// HBM time per launch against 2-5 us of MFMA time.  On the generic tiles (conv_mfma_h16.hip) every workgroup lives
// for one tile: it waits a memory latency for its first slice, re-fetches the weight slab, computes for 0.3 us and
// spends 1-2 us in an epilogue during which it has nothing in flight -- 2.0-3.4 TB/s on YOLOv4-L's pointwise layers
// (profiles/r02_layers_bf16.json), set by the bytes in flight per CU (see the note above dispatch_h16).
//
// Here one 8-wave workgroup per CU stays for the whole layer:
//   * the weight slab W[n0 : n0+BN][Cin] (<= 64 KB) is fetched ONCE into LDS;
//   * every wave walks its own strips of 32 pixels (strip = gw, gw + NW, ...) with a PRIVATE ring of three 4-KB
//     stages (32 pixels x 64 channels) filled by LDS-DMA, continuous across strips: while a strip's epilogue runs,
//     the first stages of the wave's next strip are already on their way, so every wave keeps 8 KB in flight all
//     the time (64 KB per CU) and the only wait is a counted s_waitcnt on the wave's own DMAs -- no workgroup barrier
//     after the weights have landed;
//   * a lane owns one output channel per 32-column tile (the MFMA's C layout), so the per-channel affine is four
//     registers, the BatchNorm statistics of the training forward accumulate in registers over ALL of a wave's strips
//     (two atomics per channel per wave at the very end instead of two per tile), and the 16-bit outputs go out as
//     dwords after one DPP swap between the lanes of adjacent channels (pair_pack16: sub-dword stores were the
//     first version's bottleneck), 64-byte segments per pixel.
//
// Any Cin % 8 == 0 up to 256 (laid out as 64 / 128 / 256 channels, zero-padded) and Cout >= 16.
// Epilogue = the common one: affine1 -> act1 -> (+ residual, 16-bit outputs) -> (affine2 -> act2) -> store at a channel
// offset (16-bit, or fp32 for the pred maps), or (stats != nullptr) identity + statistics of the rounded outputs.
// Scattered output stays with the generic tiles.  (The residual form serves the training step: the data gradient of a
// Bottleneck's first conv adds the shortcut's gradient, train_ops.GradSink.)
#include "conv_h16_common.h"

namespace yv4 {

constexpr int kWsWaves = 8;
constexpr int kWsThreads = kWsWaves * 64;
constexpr int kWsStages = 3;
constexpr int kWsStageBytes = 4096;   // 32 pixels x 64 channels x 2 bytes
constexpr int kWsGrid = 256;          // one workgroup per CU

__device__ __forceinline__ void act_row16_h(float (&v)[16], int act, float slope) {
  switch (act) {
    case YV4_ACT_MISH:
      mish_fast_row(v);
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

// NT: 32-column tiles of the weight slab (BN = 32 * NT output channels per workgroup)
template <bool BF16, int NT>
__global__ __launch_bounds__(kWsThreads, 1) void conv1x1_ws_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes, int ncol,
                                                                   int nstrips, int cpr_shift) {
  typedef typename Elem<BF16>::T T;
  typedef typename Elem<BF16>::V8 V8;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int BN = NT * 32;
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem_ws[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;
  // The LDS images are laid out for Cin rounded up to 64 / 128 / 256 channels; the chunks beyond the real Cin are
  // zero-filled by out-of-range DMA offsets (Cin = 32: half of every stage and of every weight row is padding --
  // these layers are HBM-bound, the idle half of the MFMA K steps costs nothing that shows)
  const int cpr = 1 << cpr_shift;       // 16-byte chunks per weight row in LDS
  const int kc_n = cpr >> 3;            // 64-channel stages per strip
  const int wpitch = cpr * 16;
  const int cin_chunks = p.Cin >> 3;    // real chunks per row

  char* Ws = smem_ws;                                                   // [BN][Cin] 16-bit, chunks XOR-swizzled
  char* ring = smem_ws + BN * wpitch + wave * (kWsStages * kWsStageBytes);
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_ws;
  const unsigned ring_lds = lds_base + (unsigned)(BN * wpitch + wave * (kWsStages * kWsStageBytes));

  // workgroup -> (column slab, walker).  Workgroups are dealt round-robin to the 8 XCDs: the `ncol` slabs of one
  // walker sit on the same XCD and read the same strips at about the same time (the second reader hits that L2).
  const unsigned b = blockIdx.x;
  const int xcd = (int)(b & 7u), local = (int)(b >> 3);
  const int col = local % ncol;
  const int walker = (local / ncol) * 8 + xcd;
  const int nwalkers = ((int)(gridDim.x >> 3) / ncol) * 8;
  const int NW = nwalkers * kWsWaves;
  const int gw = walker * kWsWaves + wave;
  const int n0 = col * BN;

  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc_h(p.w, w_bytes);

  // ---- the weight slab, once.  Row pitch Cin*2 bytes; a 16-lane pass of ds_read_b128 must touch 16 different
  // 16-byte bank groups: rows of 128 B use the (row>>1)&7 swizzle of the generic kernel, wider rows row&15.
  {
    const int groups = (BN * cpr) >> 6;   // 1-KB DMA instructions
    for (int g = wave; g < groups; g += kWsWaves) {
      const int c = g * 64 + lane;
      const int row = c >> cpr_shift;
      const int pch = c & (cpr - 1);
      const int swz = cpr >= 16 ? (row & 15) : ((row >> 1) & 7);
      const int co = n0 + row;
      const int lch = pch ^ swz;
      const unsigned voff = (co < p.Cout && lch < cin_chunks) ? (unsigned)(((int64_t)co * p.Kw + lch * 8) * 2) : kOOB;
      lds_dma16_h(rsB, lds_base + (unsigned)(g * 1024), voff, 0u);
    }
  }

  // per-channel affine of this lane's NT output channels
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

  // fragment read offsets.  A operand = the strip (row r = pixel, 128-byte rows), B operand = the slab (row = channel).
  const unsigned a_rd = (unsigned)(r * 128 + ((h ^ ((r >> 1) & 7)) << 4));
  unsigned w_rd[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = t * 32 + r;
    const int swz = cpr >= 16 ? (row & 15) : ((row >> 1) & 7);
    w_rd[t] = (unsigned)(row * wpitch + ((h ^ swz) << 4));
  }

  // stage DMA: lane fills row 8j + lane/8, physical chunk lane%8 of the slot; the logical chunk depends on j's parity
  const int my_n = gw < nstrips ? (nstrips - gw + NW - 1) / NW : 0;
  const int lrow = lane >> 3;
  const unsigned lch_even = (unsigned)(((lane & 7) ^ ((lane >> 4) & 7)) * 8);
  const unsigned lch_odd = (unsigned)(((lane & 7) ^ (((lane >> 4) + 4) & 7)) * 8);
  int iss_i = 0, iss_kc = 0, iss_slot = 0;
#define YV4_WS_ISSUE()                                                                                      \
  {                                                                                                         \
    const int row0_ = (gw + iss_i * NW) * 32 + lrow;                                                        \
    const bool live_ = iss_i < my_n;                                                                        \
    const unsigned lds_ = ring_lds + (unsigned)(iss_slot * kWsStageBytes);                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
      const int row_ = row0_ + 8 * j;                                                                       \
      const unsigned lch_ = (j & 1) ? lch_odd : lch_even;                                                   \
      const unsigned voff_ = (live_ && row_ < p.M && iss_kc * 64 + (int)lch_ < p.Cin)                       \
                                 ? (unsigned)(((int64_t)row_ * p.x_cs + p.x_co + iss_kc * 64 + (int)lch_) * 2) \
                                 : kOOB;                                                                    \
      lds_dma16_h(rsA, lds_ + (unsigned)(j * 1024), voff_, 0u);                                             \
    }                                                                                                       \
    iss_kc += 1;                                                                                            \
    const int wrap_ = iss_kc == kc_n ? 1 : 0;                                                               \
    iss_kc = wrap_ ? 0 : iss_kc;                                                                            \
    iss_i += wrap_;                                                                                         \
    iss_slot = iss_slot + 1 == kWsStages ? 0 : iss_slot + 1;                                                \
  }

  // the first two stages leave together with the weight slab (one memory latency instead of two); then the weights
  // have landed (this wave's part), then everybody's
  YV4_WS_ISSUE();
  YV4_WS_ISSUE();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float st_su[NT], st_sq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { st_su[t] = 0.f; st_sq[t] = 0.f; }

  int rslot = 0;
  for (int i = 0; i < my_n; ++i) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    for (int kc = 0; kc < kc_n; ++kc) {
      // slot (q+2) % 3 was read one iteration ago; its fragments have been consumed by issued MFMAs
      YV4_WS_ISSUE();
      // The first two stages of a strip were confirmed by the wait in front of the previous strip's stores (below);
      // a later stage has landed when at most the two younger stages (8 DMA instructions) are outstanding -- by then
      // the previous strip's stores, which count in vmcnt too, are long complete.  (A counted wait right after the
      // stores made every strip wait for their acknowledgement: 1-2 us per strip.)
      if (kc >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      const char* st = ring + rslot * kWsStageBytes;
      const unsigned kx = (unsigned)(kc << 7);   // (kc * 8) << 4: chunk index of the weight row
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const V8 fa = *reinterpret_cast<const V8*>(st + (a_rd ^ (unsigned)(s << 5)));
        V8 fb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = *reinterpret_cast<const V8*>(Ws + ((w_rd[t] ^ (unsigned)(s << 5)) ^ kx));
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = Elem<BF16>::mfma(fa, fb[t], acc[t]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      rslot = rslot + 1 == kWsStages ? 0 : rslot + 1;
    }

    // ---- epilogue of the strip: lane (r, h) holds channel n0 + 32t + r of pixels m0 + (e&3) + 8(e>>2) + 4h
    const int m0 = (gw + i * NW) * 32;
    const bool full = m0 + 32 <= p.M;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = n0 + t * 32 + r;
      if (c >= p.Cout) continue;
      float v[16];
      if (p.stats) {
        // identity epilogue; the statistics are those of the STORED (rounded) values, as in the generic kernel
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = (float)(T)acc[t][e];
          const bool in = full || (m0 + (e & 3) + 8 * (e >> 2) + 4 * h < p.M);
          st_su[t] += in ? v[e] : 0.f;
          st_sq[t] += in ? v[e] * v[e] : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = acc[t][e] * s1[t] + t1[t];
        act_row16_h(v, p.act1, p.slope1);
        if (p.res) {
          // residual (16-bit outputs only): the lane reads the dwords (two channels) of the rows it will STORE below,
          // its pair partner's dwords supply the rows the partner stores (the exchange of conv3x3_small_h16.hip)
          const bool oddr = r & 1;
          const int rbr = 4 * h + (oddr ? 16 : 0);
          const T* rp = reinterpret_cast<const T*>(p.res) + ((int64_t)(m0 + rbr) * p.r_cs + p.r_co + (c & ~1));
          unsigned mine[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int row = (j & 3) + 8 * (j >> 2);
            mine[j] = (full || m0 + rbr + row < p.M) ? *reinterpret_cast<const unsigned*>(rp + (int64_t)row * p.r_cs) : 0u;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned theirs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine[j], 0xB1, 0xF, 0xF, false);
            const unsigned a = oddr ? (theirs >> 16) : (mine[j] & 0xffffu);
            const unsigned b = oddr ? (mine[j] >> 16) : (theirs & 0xffffu);
            const unsigned short ua = (unsigned short)a, ub = (unsigned short)b;
            v[j] += (float)__builtin_bit_cast(T, ua);
            v[j + 8] += (float)__builtin_bit_cast(T, ub);
          }
        }
        if (has2) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = v[e] * s2[t] + t2[t];
          act_row16_h(v, p.act2, p.slope2);
        }
      }
      if (p.out_f32) {
        // fp32 output (the pred maps feeding decode): a lane owns one channel, a store instruction writes two runs of
        // 32 consecutive floats -- coalesced whatever the row pitch (255 channels) is
        if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* yf = reinterpret_cast<float*>(p.y) + ((int64_t)(m0 + 4 * h) * p.y_cs + p.y_co + c);
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (full || m0 + (e & 3) + 8 * (e >> 2) + 4 * h < p.M) yf[(int64_t)((e & 3) + 8 * (e >> 2)) * p.y_cs] = v[e];
        continue;
      }
      // dword stores: the even lane of a channel pair takes rows 0-3 / 8-11 (+4h), the odd lane rows 16-19 / 24-27
      const bool odd = r & 1;
      unsigned pk[8];
      pair_pack16<T>(v, odd, pk);
      // the two stages in flight (the next strip's first ones) have had this strip's last MFMAs and the epilogue
      // arithmetic to land: confirm them here, BEFORE the stores join the counter
      if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int rb = 4 * h + (odd ? 16 : 0);
      T* yb = reinterpret_cast<T*>(p.y) + ((int64_t)(m0 + rb) * p.y_cs + p.y_co + (c & ~1));
      if (full) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<unsigned*>(yb + (int64_t)((j & 3) + 8 * (j >> 2)) * p.y_cs) = pk[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (m0 + rb + (j & 3) + 8 * (j >> 2) < p.M) *reinterpret_cast<unsigned*>(yb + (int64_t)((j & 3) + 8 * (j >> 2)) * p.y_cs) = pk[j];
      }
    }
  }
#undef YV4_WS_ISSUE
  // the tail's out-of-range stage DMAs still write (zeros) into this wave's ring: drain before the LDS goes away
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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

// Slab width: the widest of 128 / 64 / 32 columns whose Cin-deep slab fits 64 KB beside the rings
static int ws_cin_lds(int Cin) { return Cin <= 64 ? 64 : (Cin <= 128 ? 128 : 256); }

static int ws_slab_cols(const ConvArgsH& a) {
  const int cout32 = (a.Cout + 31) / 32 * 32;
  for (int bn = 128; bn >= 32; bn >>= 1) {
    if (bn > cout32) continue;
    if ((long long)bn * ws_cin_lds(a.Cin) * 2 + kWsWaves * kWsStages * kWsStageBytes > 160 * 1024) continue;
    const int ncol = (a.Cout + bn - 1) / bn;
    if (32 % ncol != 0) continue;
    return bn;
  }
  return 0;
}

// Is this layer in the kernel's domain?
bool conv1x1_ws_applies(const ConvArgsH& a) {
  const bool store_ok = a.out_f32 ? a.stats == nullptr : ((a.Cout & 1) == 0 && ((a.y_cs | a.y_co) & 1) == 0);
  const bool res_ok = a.res == nullptr || (!a.out_f32 && a.stats == nullptr && ((a.r_cs | a.r_co) & 1) == 0);
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && !a.ys_on && res_ok && a.Cin >= 16 &&
         a.Cin <= 256 && (a.Cin & 7) == 0 && a.Kw == a.Cin && a.Cout >= 16 && store_ok && ws_slab_cols(a) > 0;
}

template <bool BF16, int NT>
static int launch_ws(const ConvArgsH& a, hipStream_t stream) {
  constexpr int BN = NT * 32;
  const int ncol = (a.Cout + BN - 1) / BN;
  const int cin_lds = ws_cin_lds(a.Cin);
  const size_t lds = (size_t)BN * cin_lds * 2 + (size_t)kWsWaves * kWsStages * kWsStageBytes;
  const int nstrips = (a.M + 31) / 32;
  int cpr_shift = 0;
  while ((8 << cpr_shift) < cin_lds) ++cpr_shift;
  const long long xb = (long long)a.N * a.H * a.W * a.x_cs * 2, wb = (long long)a.Cout * a.Kw * 2;
  auto kern = conv1x1_ws_kernel<BF16, NT>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024, "conv1x1_ws_h16")) return rc;
  hipLaunchKernelGGL(kern, dim3(kWsGrid), dim3(kWsThreads), lds, stream, a, (unsigned)xb, (unsigned)wb, ncol, nstrips,
                     cpr_shift);
  YV4_CHECK_LAUNCH("conv1x1_ws_h16");
  return YV4_OK;
}

int conv1x1_ws_launch(const ConvArgsH& a, bool bf16, hipStream_t s) {
  switch (ws_slab_cols(a)) {
    case 128: return bf16 ? launch_ws<true, 4>(a, s) : launch_ws<false, 4>(a, s);
    case 64: return bf16 ? launch_ws<true, 2>(a, s) : launch_ws<false, 2>(a, s);
    case 32: return bf16 ? launch_ws<true, 1>(a, s) : launch_ws<false, 1>(a, s);
    default: break;
  }
  set_error("conv1x1 ws: no weight slab of this layer fits the LDS");
  return YV4_E_UNSUPPORTED;
}

}  // namespace yv4
