// 3x3 / stride-1 / pad-1 fused convolution for gfx950 with 16-bit operands -- WIDE wave tiles on v_mfma_f32_16x16x32
// (round 4; tile id YV4_HTILE_W3x3).  The layers: the bottleneck 3x3 convs of CSPDarknet53 / PAN
// (mmdet/models/backbones/darknetcsp.py:38-64, mmdet/models/necks/yolo_neck_csp.py:11-238) and their data gradients.
//
// Why another 3x3 kernel.  conv3x3_pp_h16.hip (256 pixels x 128 channels, wave tile 64 x 64 on 32x32x16 MFMAs, two wave
// groups a phase apart) spends more issue time feeding the matrix pipe than the pipe spends computing: one ds_read_b128
// per 32 pipe cycles, ~40 % of the clocked matrix rate inside its K loop.  This kernel takes the shape
// the guide's 256 x 256 GEMM template uses: a wave owns 16 PT pixels x 64 channels (PT = 8: 128 x 64) as PT x 4
// accumulator tiles of 16 x 16, so a 64-deep K tile is 2 PT x 4 MFMAs of 16 pipe cycles fed by 2 PT + 8 fragment
// reads (0.75 reads per 32 pipe cycles at PT = 8), and its LDS-DMA fill per FLOP halves (the weight tile of a tap is
// shared by twice the pixels).  What it keeps from the ping-pong kernel: the three kw taps of a (64-channel chunk, kh)
// group read ONE LDS image of the BM + 2 source pixels (fragment row = output row + kw, image borders masked by
// redirecting a lane's read to a zero row), out-of-range DMA offsets as the zero padding, a persistent grid whose
// issue side runs on across tiles.
//
// Operand roles.  The MFMA's A operand (16 rows) is the WEIGHT tile, its B operand (16 columns) the pixels, so a lane
// of the 16 x 16 result holds 4 consecutive ROWS = output channels of ONE pixel.  The 16 MFMA rows of channel tile t
// are mapped to the wave's channels 16 (i >> 2) + 4 t + (i & 3): over t = 0..3 a lane (fq = lane >> 4) then owns the 16
// CONSECUTIVE channels 16 fq .. 16 fq + 15 of its pixel -- two 16-byte stores, a full 128-byte line per pixel and wave,
// no LDS transposition and no lane exchange in the epilogue.
//
// Pipeline.  Tiles are square in work, not in shape: BM = 16 PT WAVES_M pixels x BN = 64 (8 / WAVES_M) channels, chosen
// per layer so that the tile count fills whole rounds of CUs (256 x 256 leaves 29 % of the chip idle on a 181-tile
// layer; 192 x 256 makes it 241 tiles).  Per K tile (one tap of one chunk) every wave issues the NEXT tap's weight
// pieces (two LDS slots) and, when the tap opens a (chunk, kh) group, the next group's pixel image (two images); its
// own DMAs are confirmed by a counted s_waitcnt vmcnt at the end of the K tile, then ONE workgroup barrier publishes
// them -- a buffer is re-filled only in the K tile after the barrier that followed its last reads, a staged buffer is
// read only after the barrier that followed its wait.  Inside a K tile there is no barrier: the two waves of a SIMD drift
// apart by themselves, one reading fragments while the other's MFMAs occupy the pipe.
#include "conv_h16_common.h"
#include "conv_wide_common.h"

namespace yv4 {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool BF16> struct Mfma16;
template <> struct Mfma16<true> {
  static __device__ __forceinline__ f32x4v run(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<false> {
  static __device__ __forceinline__ f32x4v run(f16x8 a, f16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};


template <bool BF16, int PT, int WAVES_M>
__global__ __launch_bounds__(kWideThreads, 2) void conv3x3_wide_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef typename Elem<BF16>::V8 V8;
  typedef WideGeom<PT, WAVES_M, true> G_;
  constexpr int WAVES_N = G_::WAVES_N, BN = G_::BN, BM = G_::BM, WMr = G_::WMr, QA = G_::QA, PB = G_::PB;
  constexpr int PH = PT / 2;                 // pixel tiles per half
  constexpr int kRowB = 128;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_w3[];
  char* As = smem_w3;                        // [2][ARows][128 B]
  char* Bs = smem_w3 + 2 * G_::ABytes;       // [2][BN][128 B]
  float* aff = reinterpret_cast<float*>(smem_w3 + G_::RingBytes);   // [s1 | t1 | s2 | t2] x Cout

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int fr = lane & 15;
  const int fq = lane >> 4;
  // Pixel of MFMA column fr inside its 16-pixel tile.  A ds_read_b128 is served in four groups of 16 lanes that are NOT
  // contiguous ({0-3, 12-15, 20-27}, ...): a group is eight lanes of one fq reading chunk q and eight of the next reading
  // chunk q ^ 1.  With pixel = column, the kw = 1 and kw = 2 taps (rows shifted by one and two against the swizzle's
  // row pairs) put two lanes of a group on one 16-byte slot: SQ_LDS_BANK_CONFLICT was 32 % of the LDS-active cycles.
  // Columns {0-3, 12-15} take the EVEN pixels and {4-11} the odd ones: the eight lanes that read one chunk then sit on
  // rows of one parity -- one half of the banks, eight consecutive row pairs, eight different slots -- at every shift.
  const int pr = fr < 4 ? 2 * fr : (fr >= 12 ? 2 * fr - 16 : 2 * fr - 7);

  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc_h(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_w3;
  const int NHW = p.N * p.H * p.W;
  const int nwg = (int)gridDim.x;

  // virtual tile -> tile: each XCD (workgroups with equal id mod 8) walks a contiguous run of tiles, tile_n fastest
  const unsigned q8 = (unsigned)ntiles >> 3, rem8 = (unsigned)ntiles & 7u;
  auto tile_of = [&](int vt) -> unsigned {
    const unsigned x = (unsigned)vt & 7u;
    return (x < rem8 ? x * (q8 + 1) : rem8 * (q8 + 1) + (x - rem8) * q8) + ((unsigned)vt >> 3);
  };

  // ---- staging lanes: a DMA instruction of a wave fills 8 LDS rows (lane / 8) x 8 chunks (lane % 8) ----
  const int srow = 8 * wave + (lane >> 3);               // 0..63, + 64 per pass
  const int pc = lane & 7;
  const int lcA = pc ^ ((srow >> 1) & 7);                // invariant under row + 64 q
  const int lcB = pc ^ wide_swz_b(srow);                   // likewise
  int a_s[QA];                                           // source pixel of LDS row (srow + 64 q) for kh = 1, NEXT group's tile
  unsigned a_off[QA];
  unsigned b_cur[PB], b_nxt[PB];                         // weight row offsets: current K tile's tile / next group's tile
  auto issue_tile_setup = [&](int vt) {
    const bool live = vt < ntiles;
    const unsigned tile = live ? tile_of(vt) : 0u;
    const int tn = (int)(tile % (unsigned)p.tiles_n);
    const int m0i = (int)(tile / (unsigned)p.tiles_n) * BM;
    const int n0i = tn * BN;
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int row = srow + 64 * q;
      a_s[q] = (live && row < BM + 2) ? m0i - 1 + row : (int)0x40000000;     // beyond the image for every kh: zero
      a_off[q] = (unsigned)((((int64_t)(m0i - 1 + row)) * p.x_cs + p.x_co + lcA * 8) * 2);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int co = n0i + srow + 64 * q;
      b_nxt[q] = (live && co < p.Cout) ? (unsigned)(((int64_t)co * p.Kw + lcB * 8) * 2) : kOOB;
    }
  };

  // ---- fragment read addresses (tile-independent) ----
  unsigned a_rd[3][2];                       // pixel fragments: tap kw, k step; + pt * 2048 per pixel tile
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int row = wm * WMr + pr + kw;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_rd[kw][ks] = (unsigned)(row * kRowB + (((fq + 4 * ks) ^ ((row >> 1) & 7)) << 4));
  }
  const unsigned zero_rd = (unsigned)(G_::ZeroRow * kRowB);
  unsigned w_rd[2];                          // weight fragments: k step; + t * 512 per channel tile
  {
    const int row = wn * 64 + 16 * (fr >> 2) + (fr & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w_rd[ks] = (unsigned)(row * kRowB + (((fq + 4 * ks) ^ wide_swz_b(row)) << 4));
  }

  const int nchunks = p.Cin >> 6;
  const int G = 3 * nchunks;                 // (chunk, kh) groups per tile

  // ---- the layer's affine into LDS, once per workgroup ----
  const bool has2 = p.s2 != nullptr;
  for (int c = tid; c < p.Cout; c += kWideThreads) {
    aff[c] = p.s1[c];
    aff[p.Cout + c] = p.t1[c];
    aff[2 * p.Cout + c] = has2 ? p.s2[c] : 1.f;
    aff[3 * p.Cout + c] = has2 ? p.t2[c] : 0.f;
  }

#define YV4_W3_ISSUE_B(SLOT, BOFF, KB)                                                              \
  {                                                                                                 \
    const unsigned lb_ = lds_base + (unsigned)(2 * G_::ABytes + (SLOT) * G_::BBytes + 8 * wave * kRowB); \
    _Pragma("unroll") for (int q = 0; q < PB; ++q)                                                  \
        lds_dma16_h(rsB, lb_ + 64 * q * kRowB, BOFF[q], (KB));   /* (the range check sees voffset only) */ \
  }
#define YV4_W3_ISSUE_A(ABUF, Q0, Q1, KH, C0)                                                        \
  {                                                                                                 \
    const unsigned la_ = lds_base + (unsigned)((ABUF) * G_::ABytes + 8 * wave * kRowB);              \
    const int ds_ = ((KH) - 1) * p.W;                                                               \
    const unsigned step_ = (unsigned)(((int64_t)ds_ * p.x_cs + (C0)) * 2);                          \
    _Pragma("unroll") for (int q = (Q0); q < (Q1); ++q) {                                           \
      const bool ok_ = (unsigned)(a_s[q] + ds_) < (unsigned)NHW;                                    \
      lds_dma16_h(rsA, la_ + 64 * q * kRowB, ok_ ? a_off[q] + step_ : kOOB, 0u);                     \
    }                                                                                               \
  }

  // ---- prologue: group 0's image and tap 0's weights of the first tile; the issue side then points at group 1 ----
  int n_vt = (int)blockIdx.x;                // tile of the NEXT group (issue side)
  issue_tile_setup(n_vt);
#pragma unroll
  for (int q = 0; q < PB; ++q) b_cur[q] = b_nxt[q];
  YV4_W3_ISSUE_B(0, b_cur, 0u);
  YV4_W3_ISSUE_A(0, 0, QA, 0, 0);
  int n_g = 1, n_kh = 1, n_c0 = 0;           // G >= 3: group 1 is in the same tile
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // (also publishes the affine)

  unsigned T_ = 0u, GG = 0u;                 // global K-tile / group counters: weight slot T_ & 1, image GG & 1
  for (int vt = (int)blockIdx.x; vt < ntiles; vt += nwg) {
    const unsigned tile = tile_of(vt);
    const int tile_n = (int)(tile % (unsigned)p.tiles_n);
    const int tile_m = (int)(tile / (unsigned)p.tiles_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    unsigned mask9[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int m = m0 + wm * WMr + 16 * pt + pr;
      unsigned mk = 0u;
      if (m < p.M) {
        const int hw = p.H * p.W;
        const int n = fd_div(m, p.fd_hw);
        const int rm = m - n * hw;
        const int ho = fd_div(rm, p.fd_wo);
        const int wo = rm - ho * p.W;
        mk = (unsigned)tap_mask(ho - 1, wo - 1, 3, 3, p.H, p.W);
      }
      mask9[pt] = mk;
    }
    f32x4v acc[PT][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[pt][t] = f32x4v{0.f, 0.f, 0.f, 0.f};

    int c0 = 0, kh = 0;
    for (int g = 0; g < G; ++g) {
      const unsigned ab = GG & 1u;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const unsigned slot = T_ & 1u;
        const char* as_ = As + ab * G_::ABytes;
        const char* bs_ = Bs + slot * G_::BBytes;
        const int tapbit = 3 * kh + kw;
        // ---- DMA of the next K tile's weights (other slot) and, at kw == 0, of the next group's image (other image)
        // (measurement build, YV4_H16_ABLATE: 1 no weight DMA in the loop, 128 no image DMA, 2 no MFMA, 4 no barrier -- wrong
        // results on purpose, to time the loop without one of its parts)
        if (!YV4_ABLATE(p.ablate, 1)) {
          if (kw < 2) {
            YV4_W3_ISSUE_B(slot ^ 1u, b_cur, (unsigned)((((kh * 3 + kw + 1) * p.Cin) + c0) * 2));
          } else {
            YV4_W3_ISSUE_B(slot ^ 1u, b_nxt, (unsigned)((((n_kh * 3) * p.Cin) + n_c0) * 2));
          }
        }
        V8 wf[4][2] = {}, pf[PH][2] = {};
        const bool rd_ = !YV4_ABLATE(p.ablate, 8);
        // ---- phase 1: weights of channel tiles 0, 1, pixels of the first half
        if (rd_) {
        // (all four channel tiles' weight fragments now: they stay in registers for phases 2-4 anyway, and phase 2 then
        // starts its MFMAs without an LDS round trip)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const V8*>(bs_ + w_rd[ks] + t * 512);
#pragma unroll
        for (int i = 0; i < PH; ++i) {
          const bool ok = (mask9[i] >> tapbit) & 1u;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            pf[i][ks] = *reinterpret_cast<const V8*>(as_ + (ok ? a_rd[kw][ks] + (unsigned)(i * 2048) : zero_rd));
        }
        }
        if (kw == 0 && !YV4_ABLATE(p.ablate, 128)) YV4_W3_ISSUE_A(ab ^ 1u, 0, (QA + 1) / 2, n_kh, n_c0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < PH; ++i) if (!YV4_ABLATE(p.ablate, 2)) acc[i][t] = Mfma16<BF16>::run(wf[t][ks], pf[i][ks], acc[i][t]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2: channel tiles 2, 3 (fragments already in registers)
        if (kw == 0 && !YV4_ABLATE(p.ablate, 128)) YV4_W3_ISSUE_A(ab ^ 1u, (QA + 1) / 2, QA, n_kh, n_c0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 2; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < PH; ++i) if (!YV4_ABLATE(p.ablate, 2)) acc[i][t] = Mfma16<BF16>::run(wf[t][ks], pf[i][ks], acc[i][t]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 3: pixels of the second half
        if (rd_) {
#pragma unroll
        for (int i = 0; i < PH; ++i) {
          const bool ok = (mask9[PH + i] >> tapbit) & 1u;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            pf[i][ks] = *reinterpret_cast<const V8*>(as_ + (ok ? a_rd[kw][ks] + (unsigned)((PH + i) * 2048) : zero_rd));
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 2; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < PH; ++i) if (!YV4_ABLATE(p.ablate, 2)) acc[PH + i][t] = Mfma16<BF16>::run(wf[t][ks], pf[i][ks], acc[PH + i][t]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 4: the last quadrant (both operand sets are in registers)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < PH; ++i) if (!YV4_ABLATE(p.ablate, 2)) acc[PH + i][t] = Mfma16<BF16>::run(wf[t][ks], pf[i][ks], acc[PH + i][t]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's DMAs of the next K tile have landed (the next group's image, issued last at kw == 0, may still fly)
        if (kw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!YV4_ABLATE(p.ablate, 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        T_ += 1u;
      }
      // ---- group advance: current <- next; the issue side moves on by one group (possibly into the next tile) ----
      GG += 1u;
      kh = n_kh; c0 = n_c0;
#pragma unroll
      for (int q = 0; q < PB; ++q) b_cur[q] = b_nxt[q];
      n_g += 1;
      n_kh += 1;
      if (n_kh == 3) { n_kh = 0; n_c0 += kHBK; }
      if (n_g == G) {
        n_g = 0; n_c0 = 0; n_kh = 0;
        n_vt += nwg;
        issue_tile_setup(n_vt);
      }
    }
    // (kh, c0 now describe group 0 of this workgroup's next tile: reset by the assignments at the top of the loop)

    // ---- epilogue (conv_wide_common.h): lane (fr, fq) owns pixel m0 + wm WMr + 16 pt + pr, channels n0 + wn 64 + 16 fq .. + 15 ----
    if (YV4_ABLATE(p.ablate, 16)) {           // (measurement: no epilogue; keep the accumulators live)
      if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.y)[0] = acc[PT - 1][3][3] + acc[1][1][1];
      c0 = 0; kh = 0;
      continue;
    }
    wide_epilogue_h16<BF16, PT, false>(p, aff, has2, acc, m0 + wm * WMr + pr, n0 + wn * 64 + 16 * fq, lane,
                                       (unsigned)(tile_m * WAVES_M + wm));
    c0 = 0; kh = 0;
  }
#undef YV4_W3_ISSUE_A
#undef YV4_W3_ISSUE_B
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-filling tail DMAs must land before the LDS is released
}

// Is this layer in the kernel's domain?  3x3 / stride 1 / pad 1, 64-channel chunks of input, Cout in whole 16-channel
// groups with 16-byte aligned output / residual views, 16-bit output, the layer's affine beside the ring in LDS.
bool conv3x3_wide_h16_applies(const ConvArgsH& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && (a.Cin & 63) == 0 &&
         a.Kw == 9 * a.Cin && !a.ys_on && !a.out_f32 && a.Cout >= 64 && (a.Cout & 15) == 0 &&
         ((a.y_cs | a.y_co) & 7) == 0 && (a.res == nullptr || ((a.r_cs | a.r_co) & 7) == 0) && a.Cout <= 1024 &&
         a.ksplit <= 1;
}

template <bool BF16, int PT, int WAVES_M>
static int launch_w3(const ConvArgsH& a, hipStream_t stream) {
  static LdsAttrOnce once;
  return wide_launch<WideGeom<PT, WAVES_M, true>>(conv3x3_wide_h16_kernel<BF16, PT, WAVES_M>, once, "conv3x3_wide_h16", a, 2, stream);
}

// Tile shape per layer and launch: conv_wide_common.h.  pick() returns an index into kWideShapes, or -1 when no shape
// fits (LDS) -- `shape` >= 0 forces one (measurement / tests).
int conv3x3_wide_h16_pick(const ConvArgsH& a, double* rounds_eff) { return wide_pick(a, true, true, rounds_eff); }

int conv3x3_wide_h16_launch(const ConvArgsH& a, bool bf16, int shape, hipStream_t s) {
  if (shape < 0) shape = conv3x3_wide_h16_pick(a, nullptr);
  if (!wide_shape_fits("conv3x3_wide_h16", true, shape, a.Cout)) return YV4_E_UNSUPPORTED;
#define YV4_W3_CASE(I, PT_, WM_) case I: return bf16 ? launch_w3<true, PT_, WM_>(a, s) : launch_w3<false, PT_, WM_>(a, s);
  switch (shape) {
    YV4_W3_CASE(0, 8, 2)
    YV4_W3_CASE(1, 6, 2)
    YV4_W3_CASE(2, 4, 2)
    YV4_W3_CASE(3, 6, 4)
    YV4_W3_CASE(4, 4, 4)
    default: break;
  }
#undef YV4_W3_CASE
  return YV4_E_INVALID;
}

}  // namespace yv4
