// 3x3 / stride-1 / pad-1 fused convolution for gfx950 in fp32 on WIDE wave tiles (v_mfma_f32_16x16x4_f32): the fp32 form
// of conv3x3_wide_h16.hip (round 4; tile id YV4_TILE_W3x3) for the MFMA-bound layers of the headline configuration
// (mmdet/models/backbones/darknetcsp.py:38-64 bottleneck 3x3 convs, mmdet/models/necks/yolo_neck_csp.py:11-238).
//
// Same structure: a wave owns 16 PT pixels x 64 channels as PT x 4 accumulator tiles of 16 x 16, the weight tile is the
// MFMA's A operand with the channel permutation that leaves a lane 16 consecutive channels of one pixel (four 16-byte
// stores), the three kw taps of a (32-channel chunk, kh) group read one LDS image of the BM + 2 source pixels, one
// workgroup barrier per K tile, persistent grid, tile shapes chosen per layer to fill whole rounds of CUs.  What
// changes with the element type: an LDS row of 128 bytes is a 32-channel chunk; a lane's ds_read_b128 (chunk q + 4 ks of
// its row) holds FOUR K values that feed four consecutive 16x16x4 MFMAs -- MFMA step (ks, j) sums k = 16 ks + 4 q + j over
// the four lane groups q, the same four K values on both operands, so no operand needs another read -- and a K tile is
// 8 PT x 4 MFMAs of 32 pipe cycles: 8x the matrix time of the 16-bit kernel per byte staged, i.e. the loop is matrix-bound
// with the LDS / DMA / barrier work far in the shade.  The MFMAs of a phase walk the K value OUTERMOST and the 2 x PT / 2
// accumulator tiles inside: a 16x16x4 fp32 MFMA issues every 32 cycles but its result is ready after 40, so four
// back-to-back MFMAs on ONE accumulator (the natural order of a float4 fragment) would stall 8 cycles each.  One
// accumulator set: a 16x16x4 MFMA adds four products per
// rounding, so the K = 4 608 chain has the 1 152 roundings of the 32x32x2 kernels' two alternating sets.
// NOT bit-identical to the 32x32x2 tiles (fp32 products are not exact; the grouping of the K sum differs): a plan that
// must reproduce another plan's bits pins the tile id (bench.py's batch-2 check plan does).
#include "conv_f32_common.h"
#include "conv_wide_common.h"


namespace yv4 {

typedef float f32x4f __attribute__((ext_vector_type(4)));

constexpr int kF3BK = 32;          // channels per chunk = floats per 128-byte LDS row

// Diagnostic builds only (-DYV4_W3F_ABL=bits, tools/build_src_variants.sh): the K loop without one of its parts -- WRONG results
// on purpose, timing only.  1: no image pieces, 2: no weight pieces, 4: fragment reads without the border select, 8: no
// counted wait, 16: no barrier.  The product compiles with 0: every test below is a compile-time constant.
#ifndef YV4_W3F_ABL
#define YV4_W3F_ABL 0
#endif
#define W3F_ABL(BIT) ((YV4_W3F_ABL & (BIT)) != 0)

// Diagnostic build only (-DYV4_W3F_STAMP, tools/stamp_w3f.py): s_memtime sums per wave over the parts of a K tile and the
// kernel's s_memrealtime span (the clock the chip holds), read back through yv4_debug_w3f_stamps.  No stamp executes in the product.
#ifdef YV4_W3F_STAMP
__device__ unsigned long long g_w3f_stamps[1024 * 8 * 8];
#define YV4_W3F_ST(i) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); st_[i] += t__ - tl_; tl_ = t__; }
#else
#define YV4_W3F_ST(i)
#endif

template <int PT, int WAVES_M>
__global__ __launch_bounds__(kWideThreads, 2) void conv3x3_wide_f32_kernel(ConvArgs p, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef f32x4f V8;                         // one fragment read: four K values of a row
  typedef WideGeom<PT, WAVES_M, true> G_;
  constexpr int WAVES_N = G_::WAVES_N, BN = G_::BN, BM = G_::BM, WMr = G_::WMr, QA = G_::QA, PB = G_::PB;
  constexpr int PH = PT / 2;                 // pixel tiles per half
  constexpr int kRowB = 128;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_f3[];
  // smem_f3: [2][ARows][128 B] pixel images, then
  char* Bs = smem_f3 + 2 * G_::ABytes;       // [2][BN][128 B]
  float* aff = reinterpret_cast<float*>(smem_f3 + G_::RingBytes);   // [s1 | t1 | s2 | t2] x Cout

#ifdef YV4_W3F_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
  const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
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

  const u32x4_t rsA = make_rsrc(p.x, x_bytes);
  const u32x4_t rsB = make_rsrc(p.w, w_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_f3;
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
      a_off[q] = (unsigned)((((int64_t)(m0i - 1 + row)) * p.x_cs + p.x_co + lcA * 4) * 4);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int co = n0i + srow + 64 * q;
      b_nxt[q] = (live && co < p.Cout) ? (unsigned)(((int64_t)co * p.Kw + lcB * 4) * 4) : kOOB;
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
  unsigned w_rd[2];                          // weight fragments: k step; + t * 512 per channel tile
  {
    const int row = wn * 64 + 16 * (fr >> 2) + (fr & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w_rd[ks] = (unsigned)(row * kRowB + (((fq + 4 * ks) ^ wide_swz_b(row)) << 4));
  }

  const int nchunks = p.Cin >> 5;
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
        lds_dma16(rsB, lb_ + 64 * q * kRowB, BOFF[q], (KB));   /* (the range check sees voffset only) */ \
  }
#define YV4_W3_ISSUE_A(ABUF, Q0, Q1, KH, C0)                                                        \
  {                                                                                                 \
    const unsigned la_ = lds_base + (unsigned)((ABUF) * G_::ABytes + 8 * wave * kRowB);              \
    const int ds_ = ((KH) - 1) * p.W;                                                               \
    const unsigned step_ = (unsigned)(((int64_t)ds_ * p.x_cs + (C0)) * 4);                          \
    _Pragma("unroll") for (int q = (Q0); q < (Q1); ++q) {                                           \
      const bool ok_ = (unsigned)(a_s[q] + ds_) < (unsigned)NHW;                                    \
      lds_dma16(rsA, la_ + 64 * q * kRowB, ok_ ? a_off[q] + step_ : kOOB, 0u);                     \
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

  YV4_W3F_ST(0)                              // (diagnostic build: the prologue)
  unsigned T_ = 0u, GG = 0u;                 // global K-tile / group counters: weight slot T_ & 1, image GG & 1
  for (int vt = (int)blockIdx.x; vt < ntiles; vt += nwg) {
    const unsigned tile = tile_of(vt);
    const int tile_n = (int)(tile % (unsigned)p.tiles_n);
    const int tile_m = (int)(tile / (unsigned)p.tiles_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    unsigned nmask9[PT];                     // bit 3 kh + kw SET: that tap of the lane's pixel in tile pt lies outside the image
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
      nmask9[pt] = ~mk;
    }
    f32x4f acc[PT][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[pt][t] = f32x4f{0.f, 0.f, 0.f, 0.f};

    int c0 = 0, kh = 0;
    YV4_W3F_ST(5)                            // tile set-up: masks, accumulators
    for (int g = 0; g < G; ++g) {
      const unsigned ab = GG & 1u;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const unsigned slot = T_ & 1u;
        // pixel fragments: LDS address of tap kw's row for k step ks in this group's image; + 2048 per pixel tile.  A lane whose
        // tap lies outside the image reads 1 MB further on, beyond the workgroup's LDS allocation, which returns zeros
        // (wide_far_add, conv_wide_common.h): one add per read instead of a select between the row and a row of zeros
        // (profiles/r06_w3_border_select.txt: the select was 3 % of this kernel, 6-9 % of the 16-bit one)
        unsigned ard[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) ard[ks] = a_rd[kw][ks] + (lds_base + ab * (unsigned)G_::ABytes);
        const char* bs_ = Bs + slot * G_::BBytes;
        const int tapbit = 3 * kh + kw;
        // ---- DMA of the next K tile's weights (other slot) and, at kw == 0, of the next group's image (other image)
        if (W3F_ABL(2)) {
        } else if (kw < 2) {
          YV4_W3_ISSUE_B(slot ^ 1u, b_cur, (unsigned)((((kh * 3 + kw + 1) * p.Cin) + c0) * 4));
        } else {
          YV4_W3_ISSUE_B(slot ^ 1u, b_nxt, (unsigned)((((n_kh * 3) * p.Cin) + n_c0) * 4));
        }
        YV4_W3F_ST(1)                        // the weight pieces' issue (and the group advance before it)
        V8 wf[4][2], pf[PH][2];
        // ---- phase 1: weights of channel tiles 0, 1, pixels of the first half
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const V8*>(bs_ + w_rd[ks] + t * 512);
#pragma unroll
        for (int i = 0; i < PH; ++i) {
          const unsigned nb = W3F_ABL(4) ? 0u : wide_far_bit(nmask9[i], tapbit);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            pf[i][ks] = wide_lds_read<V8>((W3F_ABL(4) ? ard[ks] : wide_far_add_bit(nb, ard[ks])) + (unsigned)(i * 2048));
        }
        if (kw == 0 && !W3F_ABL(1)) YV4_W3_ISSUE_A(ab ^ 1u, 0, (QA + 1) / 2, n_kh, n_c0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j)          // K value outermost: consecutive MFMAs write DIFFERENT accumulators
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int i = 0; i < PH; ++i)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][ks][j], pf[i][ks][j], acc[i][t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 2: weights of channel tiles 2, 3
#pragma unroll
        for (int t = 2; t < 4; ++t)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) wf[t][ks] = *reinterpret_cast<const V8*>(bs_ + w_rd[ks] + t * 512);
        if (kw == 0 && !W3F_ABL(1)) YV4_W3_ISSUE_A(ab ^ 1u, (QA + 1) / 2, QA, n_kh, n_c0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j)          // K value outermost: consecutive MFMAs write DIFFERENT accumulators
#pragma unroll
            for (int t = 2; t < 4; ++t)
#pragma unroll
              for (int i = 0; i < PH; ++i)
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][ks][j], pf[i][ks][j], acc[i][t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 3: pixels of the second half
#pragma unroll
        for (int i = 0; i < PH; ++i) {
          const unsigned nb = W3F_ABL(4) ? 0u : wide_far_bit(nmask9[PH + i], tapbit);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            pf[i][ks] = wide_lds_read<V8>((W3F_ABL(4) ? ard[ks] : wide_far_add_bit(nb, ard[ks])) + (unsigned)((PH + i) * 2048));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j)          // K value outermost: consecutive MFMAs write DIFFERENT accumulators
#pragma unroll
            for (int t = 2; t < 4; ++t)
#pragma unroll
              for (int i = 0; i < PH; ++i)
                acc[PH + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][ks][j], pf[i][ks][j], acc[PH + i][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 4: the last quadrant (both operand sets are in registers)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j)          // K value outermost: consecutive MFMAs write DIFFERENT accumulators
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int i = 0; i < PH; ++i)
                acc[PH + i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][ks][j], pf[i][ks][j], acc[PH + i][t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // this wave's DMAs of the next K tile have landed (the next group's image, issued last at kw == 0, may still fly)
        YV4_W3F_ST(2)                        // the four phases: fragment reads, MFMAs, image pieces
        if (W3F_ABL(8)) {
        } else if (kw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        YV4_W3F_ST(3)                        // counted wait
        if (!W3F_ABL(16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        YV4_W3F_ST(4)                        // barrier
        T_ += 1u;
      }
      // ---- group advance: current <- next; the issue side moves on by one group (possibly into the next tile) ----
      GG += 1u;
      kh = n_kh; c0 = n_c0;
#pragma unroll
      for (int q = 0; q < PB; ++q) b_cur[q] = b_nxt[q];
      n_g += 1;
      n_kh += 1;
      if (n_kh == 3) { n_kh = 0; n_c0 += kF3BK; }
      if (n_g == G) {
        n_g = 0; n_c0 = 0; n_kh = 0;
        n_vt += nwg;
        issue_tile_setup(n_vt);
      }
    }
    // (kh, c0 now describe group 0 of this workgroup's next tile: reset by the assignments at the top of the loop)

    // ---- epilogue (conv_wide_common.h): lane (fr, fq) owns pixel m0 + wm WMr + 16 pt + pr, channels n0 + wn 64 + 16 fq .. + 15
    // (fp32: four 16-byte stores) ----
    wide_epilogue_f32<PT, false>(p, aff, has2, acc, m0 + wm * WMr + pr, n0 + wn * 64 + 16 * fq, lane,
                                 (unsigned)(tile_m * WAVES_M + wm));
    c0 = 0; kh = 0;
    YV4_W3F_ST(6)                            // epilogue
  }
#ifdef YV4_W3F_STAMP
  if (lane == 0 && blockIdx.x < 1024) {
    st_[7] = __builtin_amdgcn_s_memrealtime() - rt0_;
#pragma unroll
    for (int i = 0; i < 8; ++i) g_w3f_stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = st_[i];
  }
#endif
#undef YV4_W3_ISSUE_A
#undef YV4_W3_ISSUE_B
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-filling tail DMAs must land before the LDS is released
}

// Domain: 3x3 / stride 1 / pad 1, 32-channel chunks of input, Cout in whole 16-channel groups (64 .. 1024), 16-byte
// aligned output / residual views, no scattered output, no split-K.
bool conv3x3_wide_f32_applies(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && (a.Cin & 31) == 0 &&
         a.Kw == 9 * a.Cin && !a.ys_on && a.ksplit <= 1 && a.Cout >= 64 && (a.Cout & 15) == 0 && a.Cout <= 1024 &&
         ((a.y_cs | a.y_co) & 3) == 0 && (a.res == nullptr || ((a.r_cs | a.r_co) & 3) == 0);
}

template <int PT, int WAVES_M>
static int launch_f3(const ConvArgs& a, hipStream_t stream) {
  static LdsAttrOnce once;
  return wide_launch<WideGeom<PT, WAVES_M, true>>(conv3x3_wide_f32_kernel<PT, WAVES_M>, once, "conv3x3_wide_f32", a, 4, stream);
}

// shape choice and launch: conv_wide_common.h
int conv3x3_wide_f32_pick(const ConvArgs& a, double* rounds_eff) { return wide_pick(a, true, false, rounds_eff); }

int conv3x3_wide_f32_launch(const ConvArgs& a, int shape, hipStream_t s) {
  if (shape < 0) shape = conv3x3_wide_f32_pick(a, nullptr);
  if (!wide_shape_fits("conv3x3_wide_f32", true, shape, a.Cout)) return YV4_E_UNSUPPORTED;
  switch (shape) {
    case 0: return launch_f3<8, 2>(a, s);
    case 1: return launch_f3<6, 2>(a, s);
    case 2: return launch_f3<4, 2>(a, s);
    case 3: return launch_f3<6, 4>(a, s);
    case 4: return launch_f3<4, 4>(a, s);
    default: break;
  }
  return YV4_E_INVALID;
}

}  // namespace yv4

#ifdef YV4_W3F_STAMP
extern "C" int yv4_debug_w3f_stamps(unsigned long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const size_t bytes = sizeof(unsigned long long) * (size_t)(n < 1024 * 64 ? n : 1024 * 64);
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(yv4::g_w3f_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
