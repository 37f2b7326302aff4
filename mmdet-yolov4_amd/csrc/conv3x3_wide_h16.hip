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
// group read ONE LDS image of the BM + 2 source pixels (fragment row = output row + kw; a lane whose tap lies outside the
// image reads beyond the workgroup's LDS allocation, which returns zeros), out-of-range DMA offsets as the zero padding, a persistent grid whose
// issue side runs on across tiles.
//
// Operand roles.  The MFMA's A operand (16 rows) is the WEIGHT tile, its B operand (16 columns) the pixels, so a lane
// of the 16 x 16 result holds 4 consecutive ROWS = output channels of ONE pixel.  The 16 MFMA rows of channel tile t
// are mapped to the wave's channels 16 (i >> 2) + 4 t + (i & 3): over t = 0..3 a lane (fq = lane >> 4) then owns the 16
// CONSECUTIVE channels 16 fq .. 16 fq + 15 of its pixel -- two 16-byte stores, a full 128-byte line per pixel and wave,
// no LDS transposition and no lane exchange in the epilogue.
//
// Tiles are square in work, not in shape: BM = 16 PT WAVES_M pixels x BN = 64 (8 / WAVES_M) channels, chosen per layer by
// a time model of a round of tiles (conv_wide_common.h: wide_pick).  Two weight slots, two pixel images; a buffer is
// re-filled only after the barrier that followed its last reads, a staged buffer is read only after the barrier that
// followed its issuer's s_waitcnt.
//
// Pipeline (round 6; DESIGN 9.2 has the measurements).  The two waves of a SIMD -- w and w + 4 -- take turns: a wave
// alternates a LOAD interval (its LDS-DMA pieces; the fragment reads of SK k steps of 32 into registers; its waits) and an
// MFMA interval (SK x 4 PT MFMAs back to back from registers, one per accumulator), waves 4-7 run one interval behind waves
// 0-3, a workgroup barrier separates the intervals, and the groups are re-aligned around the epilogue (waves 4-7 enter a
// tile one barrier late, waves 0-3 leave it one barrier late).  While one wave of a SIMD is parked in its vector-memory
// issue (~100 cycles per 1 KB piece with four waves issuing: the CU takes in a piece per ~28 cycles) or waits for its
// fragments, its partner owns the matrix pipe.  SK = 1 at eight pixel tiles per wave (the fragments of one k step are 48
// registers beside 128 accumulators), 2 below.
// Issue roles (the SK = 2 shapes): waves 0-3 issue every weight piece and confirm them at the end of their MFMA interval,
// waves 4-7 issue every piece of the next (chunk, kh) group's pixel image and confirm them ONCE per group -- a wave's vmcnt
// is one in-order counter, so a wave that confirmed weight pieces every K tile would give every older image piece one K
// tile to land, whatever the schedule says.
// Per-piece offsets are a lane constant + scalar arithmetic (no per-piece registers), the border masks are two registers,
// the fragment addresses are computed where they are used (hoisted they were spilled and reloaded inside the K loop).
// Same K order and epilogue as before: the same bits (tests/test_gpu_h16.py::test_wide3x3_matches_generic_bitwise).
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


// Compile-time ablation (-DYV4_W3_ABL=bits, tools/abl_w3.sh: one library per variant, no run-time branch in any timed
// kernel -- the run-time switches of the measurement build cost this loop more than the parts they remove): 1 no weight
// DMA in the loop, 128 no image DMA, 2 no MFMA, 4 no barrier, 8 no fragment reads, 16 no epilogue, 32 fragment reads without the
// border select.  Wrong results on purpose.
#ifndef YV4_W3_ABL
#define YV4_W3_ABL 0
#endif
#define W3_ABL(BIT) ((YV4_W3_ABL & (BIT)) != 0)
#ifndef YV4_W3_SPLIT
#define YV4_W3_SPLIT 0                      // weight pieces the issuing role sends from its MFMA interval (0, or up to PT): 2 and 4 measured within 2 % of 0 (profiles/r06_w3_split_issue.txt)
#endif
#ifndef YV4_W3_ILV
#define YV4_W3_ILV 0                        // 1: the issuing role interleaves its fragment reads with its pieces (measured: nothing, profiles/r06_w3_modes3.txt)
#endif
#ifndef YV4_W3_ROLES
#define YV4_W3_ROLES 1                      // waves 0-3 issue every weight piece, waves 4-7 every image piece (SK = 2 shapes)
#endif
// Diagnostic build only (-DYV4_W3_STAMP, tools/stamp_w3.sh): s_memtime sums per wave over the parts of a K tile, read
// back through yv4_debug_w3_stamps.  No stamp executes in the product.
#ifdef YV4_W3_STAMP
__device__ unsigned long long g_w3_stamps[1024 * 8 * 8];
#define YV4_W3_ST(i) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); st_[i] += t__ - tl_; tl_ = t__; }
#else
#define YV4_W3_ST(i)
#endif

template <bool BF16, int PT, int WAVES_M>
__global__ __launch_bounds__(kWideThreads, 2) void conv3x3_wide_h16_kernel(ConvArgsH p, unsigned x_bytes, unsigned w_bytes, int ntiles) {
  typedef typename Elem<BF16>::V8 V8;
  typedef WideGeom<PT, WAVES_M, true> G_;
  constexpr int WAVES_N = G_::WAVES_N, BN = G_::BN, BM = G_::BM, WMr = G_::WMr, QA = G_::QA, PB = G_::PB;
  constexpr int QH = (QA + 1) / 2;           // image passes issued at kw = 0 (the rest at kw = 1)
#ifdef YV4_W3_SK
  constexpr int SK = YV4_W3_SK;
#else
  constexpr int SK = PT >= 8 ? 1 : 2;        // k steps (of 32) per LOAD / MFMA interval: the fragments of SK steps are in registers
#endif
  constexpr int kRowB = 128;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem_w3[];
  // smem_w3: [2][ARows][128 B] pixel images, then
  char* Bs = smem_w3 + 2 * G_::ABytes;       // [2][BN][128 B]
  float* aff = reinterpret_cast<float*>(smem_w3 + G_::RingBytes);   // [s1 | t1 | s2 | t2] x Cout

#ifdef YV4_W3_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
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

  // ---- staging lanes: a DMA instruction (one 1 KB "piece") of a wave fills 8 LDS rows (lane / 8) x 8 chunks (lane % 8):
  // rows 8 wave + lane / 8 of every 64-row pass q.  Per-piece offsets are a lane constant + scalar arithmetic.
  // ROLES (the SK = 2 shapes): waves 0-3 issue EVERY weight piece (both 32-row halves vv of a pass from issue position
  // wave & 3), waves 4-7 every image piece.  A wave's vmcnt is one in-order counter: a wave that confirms its weight pieces
  // every K tile thereby also waits for every older image piece, i.e. gives the image ONE K tile to arrive whatever the
  // schedule says -- too little when the image comes from HBM (inside a network: profiles/r06_w3_modes.txt).  With the
  // roles split, waves 4-7 wait once per (chunk, kh) group and the image has two to three K tiles.
  constexpr bool ROLES = YV4_W3_ROLES != 0 && SK == 2;
  constexpr int NV = ROLES ? 2 : 1, PR = 64 / NV;          // pieces per wave and pass; rows between them
  const bool grpB = wave >= 4;                             // second wave of its SIMD (wave-uniform)
  const bool issW = !ROLES || !grpB, issI = !ROLES || grpB;
  const int iw = ROLES ? (wave & 3) : wave;
  const int srow = 8 * iw + (lane >> 3);                   // 0..PR-1
  const int pc = lane & 7;
  const int lcA = pc ^ ((srow >> 1) & 7);                  // invariant under row + 32
  const int lcB = pc ^ wide_swz_b(srow);                   // row + 32 flips bit 2 of the swizzle, row + 64 nothing
  const unsigned a_lane = (unsigned)((srow * p.x_cs + p.x_co + lcA * 8) * 2);
  const unsigned b_lane = (unsigned)((srow * p.Kw + lcB * 8) * 2);
  const unsigned b_lane1 = (unsigned)(((srow + 32) * p.Kw + (lcB ^ 4) * 8) * 2);
  // scalars of the NEXT group's tile (issue side); sB_*: byte offset of weight row n0 (beyond the tensor when not live)
  int n_m0 = 0;
  bool n_live = false;
  unsigned sB_cur = 0u, sB_nxt = 0u;
  auto issue_tile_setup = [&](int vt) {
    n_live = vt < ntiles;
    const unsigned tile = n_live ? tile_of(vt) : 0u;
    n_m0 = (int)(tile / (unsigned)p.tiles_n) * BM;
    sB_nxt = n_live ? (tile % (unsigned)p.tiles_n) * (unsigned)(BN * p.Kw * 2) : 0xF0000000u;
  };

  // ---- fragment read addresses (tile-independent) ----
  // pixel fragments: tap kw reads LDS row arow0 + kw (+ 16 per pixel tile), k step ks its chunk fq + 4 ks under the row's
  // swizzle -- computed where it is used (six loop-invariant addresses and their 6 PT sums with 2048 pt would otherwise be
  // hoisted out of the loops and spilled by the epilogue's register pressure, to be reloaded INSIDE the K loop)
  const int arow0 = wm * WMr + pr;
#ifndef YV4_W3_HOIST
#define YV4_W3_HOIST 1
#endif
  // the six fragment row addresses (tap kw, k step) in registers where six pixel tiles per wave leave room for them (253 of
  // 256, nothing spilled -- before the border select went, 9.6d, the same six were spilled by the epilogue's pressure and
  // reloaded inside the K loop); computed in the LOAD interval otherwise.  0-3 % per layer (profiles/r06_w3_hoist.txt)
  constexpr bool HOIST = YV4_W3_HOIST != 0 && PT <= 6;
  unsigned a_rd6[3][2];
  if (HOIST) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int arow = arow0 + kw;
        a_rd6[kw][ks] = (unsigned)(arow * kRowB + (((fq + 4 * ks) ^ ((arow >> 1) & 7)) << 4)) + lds_base;
      }
  }
  unsigned w_rd[2];                          // weight fragments: k step; + t * 512 per channel tile
  {
    const int row = wn * 64 + 16 * (fr >> 2) + (fr & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w_rd[ks] = (unsigned)(row * kRowB + (((fq + 4 * ks) ^ wide_swz_b(row)) << 4));
  }

  const int nchunks = p.Cin >> 6;
  const int G = 3 * nchunks;                 // (chunk, kh) groups per tile

  const bool has2 = p.s2 != nullptr;

  // weight piece q of a K tile: rows n0 + srow + 64 q.  A row at or beyond Cout has its vector offset beyond the tensor and
  // reads as zeros (the descriptor's range check sees the vector offset only: the K offset is the soffset)
#define YV4_W3_PIECE_B(SLOT, SB, KB, q, vv)                                                          \
  lds_dma16_h(rsB, lds_base + (unsigned)(2 * G_::ABytes + (SLOT) * G_::BBytes + (8 * iw + 64 * (q) + 32 * (vv)) * kRowB), \
              ((vv) ? b_lane1 : b_lane) + ((SB) + (unsigned)(64 * (q)) * (unsigned)(p.Kw * 2)), (KB));
  // image piece q of a (chunk, kh) group: LDS row r = srow + 64 q holds source pixel m0 - 1 + r + (kh - 1) W.  Pixels outside
  // the tensor and the rows from BM + 2 on (row BM + 2 is the fragments' zero row) are zero-filled.
#define YV4_W3_PIECE_A(ABUF, KH, C0, q, vv)                                                          \
  if (!ROLES || 64 * (q) + 32 * (vv) <= BM + 2) {      /* (ROLES: a piece that starts beyond the zero row is never read) */ \
    const int sS_ = n_m0 - 1 + ((KH) - 1) * p.W;                                                    \
    const unsigned sOff_ = ((unsigned)sS_ * (unsigned)p.x_cs + (unsigned)(C0)) * 2u;                 \
    const int r0_ = 64 * (q) + 32 * (vv);                                                           \
    const bool ok_ = n_live && (r0_ + PR <= BM + 2 || srow < BM + 2 - r0_) && (unsigned)(srow + sS_ + r0_) < (unsigned)NHW; \
    lds_dma16_h(rsA, lds_base + (unsigned)((ABUF) * G_::ABytes + (8 * iw + r0_) * kRowB),            \
                ok_ ? a_lane + (sOff_ + (unsigned)r0_ * (unsigned)(p.x_cs * 2)) : kOOB, 0u);         \
  }

  // ---- prologue: group 0's image and tap 0's weights of the first tile; the issue side then points at group 1 ----
  int n_vt = (int)blockIdx.x;                // tile of the NEXT group (issue side)
  issue_tile_setup(n_vt);
  sB_cur = sB_nxt;
  if (issW) {
#pragma unroll
    for (int q = 0; q < PB * NV; ++q) YV4_W3_PIECE_B(0, sB_cur, 0u, q / NV, q % NV)
  }
  if (issI) {
#pragma unroll
    for (int q = 0; q < QA * NV; ++q) YV4_W3_PIECE_A(0, 0, 0, q / NV, q % NV)
  }
  int n_g = 1, n_kh = 1, n_c0 = 0;           // G >= 3: group 1 is in the same tile
  // ---- the layer's affine into LDS, once per workgroup -- behind the first fills' issue, so that its memory round trip
  // runs beside theirs instead of in front of them ----
  for (int c = tid; c < p.Cout; c += kWideThreads) {
    aff[c] = p.s1[c];
    aff[p.Cout + c] = p.t1[c];
    aff[2 * p.Cout + c] = has2 ? p.s2[c] : 1.f;
    aff[3 * p.Cout + c] = has2 ? p.t2[c] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();              // (also publishes the affine)

  YV4_W3_ST(0)                               // (diagnostic build: the prologue)
  unsigned T_ = 0u, GG = 0u;                 // global K-tile / group counters: weight slot T_ & 1, image GG & 1
  for (int vt = (int)blockIdx.x; vt < ntiles; vt += nwg) {
    const unsigned tile = tile_of(vt);
    const int tile_n = (int)(tile % (unsigned)p.tiles_n);
    const int tile_m = (int)(tile / (unsigned)p.tiles_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    // border masks of the wave's PT pixels: tap (kh, kw) of pixel tile pt is inside the image iff bit 3 pt + kh of rowm and
    // bit 3 pt + kw of colm are set (two registers instead of PT 9-bit masks; a pixel beyond M has no row bit)
    unsigned rowm = 0u, colm = 0u;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int m = m0 + wm * WMr + 16 * pt + pr;
      if (m < p.M) {
        const int hw = p.H * p.W;
        const int n = fd_div(m, p.fd_hw);
        const int rm = m - n * hw;
        const int ho = fd_div(rm, p.fd_wo);
        const int wo = rm - ho * p.W;
        const unsigned rb = (ho > 0 ? 1u : 0u) | 2u | (ho + 1 < p.H ? 4u : 0u);
        const unsigned cb = (wo > 0 ? 1u : 0u) | 2u | (wo + 1 < p.W ? 4u : 0u);
        rowm |= rb << (3 * pt);
        colm |= cb << (3 * pt);
      }
    }
    f32x4v acc[PT][4];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[pt][t] = f32x4v{0.f, 0.f, 0.f, 0.f};

    if (grpB) __builtin_amdgcn_s_barrier();   // waves 4-7 run one interval behind their SIMD partners
    int c0 = 0, kh = 0;
    for (int g = 0; g < G; ++g) {
      const unsigned ab = GG & 1u;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const unsigned slot = T_ & 1u;
        const char* bs_ = Bs + slot * G_::BBytes;
        const unsigned okm = (rowm >> kh) & (colm >> kw);      // bit 3 pt: tap (kh, kw) of pixel tile pt is inside
#pragma unroll
        for (int ks0 = 0; ks0 < 2; ks0 += SK) {
          // ================= LOAD interval of steps ks0 .. ks0 + SK - 1: the SIMD partner is in its MFMA interval =================
          // Order inside the interval (measured, profiles/r06_w3_order.txt): an LDS-DMA piece parks the wave in its vector-memory
          // issue for ~100 cycles (four waves issuing), a fragment read is issued in a few cycles and returns by itself.  SK = 2:
          // the weight pieces FIRST (those of waves 4-7 have to land within this interval), then the reads, then the next group's
          // image; SK = 1: the reads first (their latency runs under the pieces' issue), then the pieces.  Reads in front of
          // every piece (interleaved) lost 8 % against either.
          V8 wf[SK][4] = {}, pf[SK][PT] = {};
          unsigned ard[SK];
#pragma unroll
          for (int k = 0; k < SK; ++k) {
            int arow = arow0;
            asm volatile("" : "+v"(arow));
            arow += kw;
            ard[k] = HOIST ? a_rd6[kw][ks0 + k] + ab * (unsigned)G_::ABytes
                           : (unsigned)(arow * kRowB + (((fq + 4 * (ks0 + k)) ^ ((arow >> 1) & 7)) << 4)) + (lds_base + ab * (unsigned)G_::ABytes);
          }
          // Border taps: a lane whose tap lies outside the image reads BEYOND the workgroup's LDS allocation, which returns
          // zeros (tools/microbench/lds_oob_read.hip) -- one add per read (bit 3 pt of `nokm` shifted to 1 MB) instead of a
          // select between the pixel's row and a row of zeros.  The select was 9-14 % of this kernel: its VALU work sits in
          // the LOAD interval, the longer of the two (profiles/r06_w3_border_select.txt).
          const unsigned nokm = ~okm;
          constexpr int NR = (4 + PT) * SK;                                   // fragment reads of the interval
          const int NPW = (ks0 == 0 && !W3_ABL(1)) ? PB * NV : 0;             // its weight pieces ...
          // ... of which the last NPC go out in the MFMA interval, one behind each of its first MFMAs (YV4_W3_SPLIT: the
          // issuing role's LOAD interval is its eight pieces; four of them under its own MFMAs cost the matrix pipe a bubble
          // each but shorten the interval both SIMD partners wait for)
          const int NPC = (ROLES && YV4_W3_SPLIT && NPW > 0) ? YV4_W3_SPLIT : 0;
          const int q0i = (kw == 0 ? 0 : QH) * NV;
          const int NPI = (ks0 + SK == 2 && kw < 2 && !W3_ABL(128)) ? (kw == 0 ? QH : QA - QH) * NV : 0;   // ... and image pieces
#define YV4_W3_READ1(r)                                                                              \
          if (!W3_ABL(8)) {                                                                          \
            const int k_ = (r) / (4 + PT), j_ = (r) % (4 + PT);                                      \
            if (j_ < 4) {                                                                            \
              wf[k_][j_] = *reinterpret_cast<const V8*>(bs_ + w_rd[ks0 + k_] + j_ * 512);            \
            } else {                                                                                 \
              const unsigned ad_ = W3_ABL(32) ? ard[k_] : wide_far_add(nokm, 3 * (j_ - 4), ard[k_]);   /* (32: no border handling) */ \
              pf[k_][j_ - 4] = wide_lds_read<V8>(ad_ + (unsigned)((j_ - 4) * 2048));   /* (ard holds LDS addresses) */               \
            }                                                                                        \
          }
#define YV4_W3_WEIGHT_PIECES                                                                         \
          if (issW) {                                                                                \
            _Pragma("unroll") for (int q = 0; q < NPW - NPC; ++q) {                                  \
              if (kw < 2) YV4_W3_PIECE_B(slot ^ 1u, sB_cur, (unsigned)((((kh * 3 + kw + 1) * p.Cin) + c0) * 2), q / NV, q % NV) \
              else YV4_W3_PIECE_B(slot ^ 1u, sB_nxt, (unsigned)((((n_kh * 3) * p.Cin) + n_c0) * 2), q / NV, q % NV) \
            }                                                                                        \
          }
#if YV4_W3_ILV
          if (ROLES && issW && NPW > 0) {
            // the issuing role: a few reads behind every piece -- they return while the wave is parked in the next piece's issue
            constexpr int CH = (NR + PB * NV - 1) / (PB * NV);
#pragma unroll
            for (int q = 0; q < NPW; ++q) {
              if (kw < 2) YV4_W3_PIECE_B(slot ^ 1u, sB_cur, (unsigned)((((kh * 3 + kw + 1) * p.Cin) + c0) * 2), q / NV, q % NV)
              else YV4_W3_PIECE_B(slot ^ 1u, sB_nxt, (unsigned)((((n_kh * 3) * p.Cin) + n_c0) * 2), q / NV, q % NV)
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int r = q * CH; r < (q + 1) * CH && r < NR; ++r) YV4_W3_READ1(r)
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = NPW * CH; r < NR; ++r) YV4_W3_READ1(r)
          } else
#endif
          {
          if (SK == 2) {
            YV4_W3_WEIGHT_PIECES
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) YV4_W3_READ1(r)
          __builtin_amdgcn_sched_barrier(0);
          if (SK == 1) {
            YV4_W3_WEIGHT_PIECES
          }
          }
          if (issI) {
#pragma unroll
            for (int q = 0; q < NPI; ++q) YV4_W3_PIECE_A(ab ^ 1u, n_kh, n_c0, (q0i + q) / NV, (q0i + q) % NV)
          }
#undef YV4_W3_WEIGHT_PIECES
#undef YV4_W3_READ1
          // waves 4-7 confirm their DMAs here, waves 0-3 at the end of their MFMA interval: the same barrier publishes both.
          // In flight and allowed to stay: the image pieces issued in this K tile (kw = 0: QH passes, kw = 1: the rest).
          if (ks0 + SK == 2 && grpB) {
            if (kw == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (ROLES) {}                                   // (image pieces only: confirmed once per group)
            else if (kw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QH) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA - QH) : "memory");
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the fragments are in registers before the roles swap
          YV4_W3_ST(1)
          if (!W3_ABL(4)) __builtin_amdgcn_s_barrier();
          YV4_W3_ST(2)
          // ================= MFMA interval: SK x 4 PT MFMAs from registers =================
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int k = 0; k < SK; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int i = 0; i < PT; ++i) {
                if (!W3_ABL(2)) acc[i][t] = Mfma16<BF16>::run(wf[k][t], pf[k][i], acc[i][t]);
                else asm volatile("" ::"v"(wf[k][t]), "v"(pf[k][i]));      // (ablation: the fragment reads stay)
                if (NPC > 0 && k == 0 && t == 0 && i < NPC && issW) {
                  const int q = NPW - NPC + i;
                  __builtin_amdgcn_sched_barrier(0);
                  if (kw < 2) YV4_W3_PIECE_B(slot ^ 1u, sB_cur, (unsigned)((((kh * 3 + kw + 1) * p.Cin) + c0) * 2), q / NV, q % NV)
                  else YV4_W3_PIECE_B(slot ^ 1u, sB_nxt, (unsigned)((((n_kh * 3) * p.Cin) + n_c0) * 2), q / NV, q % NV)
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          YV4_W3_ST(3)
          if (ks0 + SK == 2 && !grpB) {
            if (ROLES || kw == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (ROLES: weight pieces only)
            else if (kw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QH) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA - QH) : "memory");
          }
          YV4_W3_ST(4)
          if (!W3_ABL(4)) __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          YV4_W3_ST(5)
        }
        T_ += 1u;
      }
      // ---- group advance: current <- next; the issue side moves on by one group (possibly into the next tile) ----
      GG += 1u;
      kh = n_kh; c0 = n_c0;
      sB_cur = sB_nxt;
      n_g += 1;
      n_kh += 1;
      if (n_kh == 3) { n_kh = 0; n_c0 += kHBK; }
      if (n_g == G) {
        n_g = 0; n_c0 = 0; n_kh = 0;
        n_vt += nwg;
        issue_tile_setup(n_vt);
      }
    }
    if (!grpB) __builtin_amdgcn_s_barrier();  // waves 0-3 wait one interval: both groups run their epilogues together
    YV4_W3_ST(6)
    // (kh, c0 now describe group 0 of this workgroup's next tile: reset by the assignments at the top of the loop)

    // ---- epilogue (conv_wide_common.h): lane (fr, fq) owns pixel m0 + wm WMr + 16 pt + pr, channels n0 + wn 64 + 16 fq .. + 15 ----
    if (W3_ABL(16)) {           // (measurement: no epilogue; keep the accumulators live)
      if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.y)[0] = acc[PT - 1][3][3] + acc[1][1][1];
      c0 = 0; kh = 0;
      continue;
    }
    wide_epilogue_h16<BF16, PT, false>(p, aff, has2, acc, m0 + wm * WMr + pr, n0 + wn * 64 + 16 * fq, lane,
                                       (unsigned)(tile_m * WAVES_M + wm));
    c0 = 0; kh = 0;
    YV4_W3_ST(7)
  }
#ifdef YV4_W3_STAMP
  if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g_w3_stamps[((size_t)blockIdx.x * 8 + wave) * 8 + i] = st_[i];
  }
#endif
#undef YV4_W3_PIECE_A
#undef YV4_W3_PIECE_B
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
int conv3x3_wide_h16_pick(const ConvArgsH& a, double* rounds_eff) { return wide_pick(a, true, true, rounds_eff, kWideShapes3x3H); }

int conv3x3_wide_h16_launch(const ConvArgsH& a, bool bf16, int shape, hipStream_t s) {
  if (shape < 0) shape = conv3x3_wide_h16_pick(a, nullptr);
  if (!wide_shape_fits("conv3x3_wide_h16", true, shape, a.Cout, kWideShapes3x3H)) return YV4_E_UNSUPPORTED;
#define YV4_W3_CASE(I, PT_, WM_) case I: return bf16 ? launch_w3<true, PT_, WM_>(a, s) : launch_w3<false, PT_, WM_>(a, s);
  switch (shape) {
    YV4_W3_CASE(0, 8, 2)
    YV4_W3_CASE(1, 6, 2)
    YV4_W3_CASE(2, 4, 2)
    YV4_W3_CASE(3, 6, 4)
    YV4_W3_CASE(4, 4, 4)
    YV4_W3_CASE(5, 3, 8)
    YV4_W3_CASE(6, 2, 8)
    default: break;
  }
#undef YV4_W3_CASE
  return YV4_E_INVALID;
}

}  // namespace yv4

#ifdef YV4_W3_STAMP
extern "C" int yv4_debug_w3_stamps(unsigned long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const size_t bytes = sizeof(unsigned long long) * (size_t)(n < 1024 * 64 ? n : 1024 * 64);
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(yv4::g_w3_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
