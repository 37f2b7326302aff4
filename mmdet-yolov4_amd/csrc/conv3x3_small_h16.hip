// 3x3 / stride-1 / pad-1 fused convolution for gfx950 with 16-bit operands and FEW channels (Cin 16, 32 or 64,
// Cout 32 or 64) -- the first bottlenecks of CSPDarknet at 1/2 and 1/4 of the input resolution (darknetcsp.py:38-64,
// Bottleneck.conv2 of the `bottleneck` / first `csp` stages; yolov4s runs them at 16 and 32 channels).
//
// Why a fourth kernel.  These layers are HBM-bound (a 16 -> 32 conv at 208 x 208 x 256 moves 1.8 GB for 52 GFLOP), but
// on the generic implicit-GEMM tiles a workgroup lives for 128 pixels: it fetches its nine im2col slices through the
// per-lane (tap, channel) decode of the Cin % 64 != 0 path, half of its 64 output columns are padding, and prologue,
// three K slices and an epilogue with a residual take 10 us -- 1.1 ms for a layer whose traffic needs 0.32 ms.
//
// One persistent 8-wave workgroup per CU walks 16 x 16 tiles of the output map:
//   * the weights (Cout x 9 Cin, <= 72 KB) are loaded into LDS once;
//   * the 18 x 18 x Cin input tile of the NEXT output tile is fetched by LDS-DMA into the second of two LDS buffers
//     while the current one is computed (out-of-image pixels arrive as zeros: the buffer descriptor's range check IS
//     the padding), so every input pixel is read from HBM 1.27 times instead of being re-gathered nine times;
//   * wave w computes output rows 2w, 2w + 1 (32 pixels) x Cout from LDS, operand reads staged one tap ahead of the
//     MFMAs that use them; epilogue = the common one (affine1 -> act1 -> +residual -> affine2 -> act2), residual loads
//     and stores as dwords of channel pairs (pair_pack16).
// One workgroup barrier per tile.
#include "conv_h16_common.h"

namespace yv4 {

// 8 waves per workgroup, or 4 (template parameter NW) with several workgroups per CU when the LDS footprint allows
constexpr int kS3T = 16;                 // output tile edge
constexpr int kS3I = kS3T + 2;           // input tile edge (18)
constexpr int kS3Pix = kS3I * kS3I;      // 324

// CINH = Cin / 16 (1, 2, 4), NT = Cout / 32 (1, 2), NW = waves per workgroup (8 or 4: then a wave takes two of the
// tile's eight row pairs and two to four workgroups share a CU and drift out of phase)
template <bool BF16, int CINH, int NT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv3x3_small_kernel(ConvArgsH p, unsigned x_bytes, int tiles_x, int tiles_y,
                                                                      int ntiles, FastDiv fd_tx, FastDiv fd_txy) {
  typedef typename Elem<BF16>::T T;
  typedef typename Elem<BF16>::V8 V8;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int Cin = CINH * 16;
  constexpr int Cout = NT * 32;
  constexpr int K = 9 * Cin;
  constexpr int WPitch = K * 2;            // bytes per weight row
  constexpr int WCpr = K / 8;              // 16-byte chunks per weight row (18 / 36 / 72)
  constexpr int Cpp = Cin / 8;             // 16-byte chunks per pixel (2 / 4 / 8)
  constexpr int PixB = Cin * 2;
  constexpr int XBytes = (kS3Pix * PixB + 1023) & ~1023;     // one input tile, rounded to whole DMA instructions
  constexpr int NGroups = XBytes / 1024;
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  constexpr int kS3Threads = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem_s3[];
  char* Xl = smem_s3;                      // [2][324 pixels][Cin], chunks XOR-swizzled with the pixel index
  char* Wl = smem_s3 + 2 * XBytes;         // [Cout][K], chunks XOR-swizzled with the row index

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_s3;
  const u32x4_t rsA = make_rsrc_h(p.x, x_bytes);

  // chunk swizzles: a 16-lane group of ds_read_b128 must touch 16 different 16-byte slots of the 256-byte bank row
  auto xswz = [](int q) { return CINH == 1 ? ((q >> 3) & 1) : (CINH == 2 ? ((q >> 2) & 3) : ((q >> 1) & 7)); };
  auto wswz = [](int row) { return CINH == 1 ? ((row >> 3) & 1) : (CINH == 2 ? ((row >> 2) & 3) : ((row >> 1) & 7)); };

  // ---- once: the weights
  for (int c = tid; c < Cout * WCpr; c += kS3Threads) {
    const int row = c / WCpr, kc = c - row * WCpr;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < p.Cout) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w) + (size_t)row * WPitch + kc * 16);
    *reinterpret_cast<uint4*>(Wl + row * WPitch + ((kc ^ wswz(row)) << 4)) = v;
  }
  // per-channel affine of this lane's output channels
  float s1[NT], t1[NT], s2[NT], t2[NT];
  const bool has2 = p.s2 != nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = t * 32 + r;
    const bool ok = c < p.Cout;
    s1[t] = ok ? p.s1[c] : 0.f;
    t1[t] = ok ? p.t1[c] : 0.f;
    s2[t] = (ok && has2) ? p.s2[c] : 1.f;
    t2[t] = (ok && has2) ? p.t2[c] : 0.f;
  }

  // input-tile DMA: instruction g of a tile fills physical chunks 64 g .. 64 g + 63 of the buffer; a lane derives
  // the (pixel, logical chunk) its slot holds
  auto issue_tile = [&](int tile, int buf) {
    const int n = fd_div(tile, fd_txy);
    const int rem = tile - n * (tiles_x * tiles_y);
    const int ty = fd_div(rem, fd_tx);
    const int tx = rem - ty * tiles_x;
    const int iy0 = ty * kS3T - 1, ix0 = tx * kS3T - 1;
    for (int g = wave; g < NGroups; g += NW) {
      const int c = g * 64 + lane;
      const int q = c / Cpp, pch = c - q * Cpp;
      const int py = q / kS3I, px = q - py * kS3I;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = q < kS3Pix && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)((((int64_t)(n * p.H + iy) * p.W + ix) * p.x_cs + p.x_co + (pch ^ xswz(q)) * 8) * 2) : kOOB;
      lds_dma16_h(rsA, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(buf * XBytes + g * 1024))), voff, 0u);
    }
  };

  // fragment read bases.  A operand: pixel (oyl + dy, oxl + dx) of the input tile; B operand: weight row 32 t + r.
  unsigned wrow[NT], whs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = t * 32 + r;
    wrow[t] = (unsigned)(row * WPitch);
    whs[t] = (unsigned)wswz(row);
  }
  const bool odd = r & 1;
  // training forward (identity epilogue): BatchNorm sums of the stored values, kept in registers over all of this
  // wave's tiles -- two atomics per channel and wave at the very end
  float st_su[NT], st_sq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { st_su[t] = 0.f; st_sq[t] = 0.f; }

  int tile = (int)blockIdx.x;
  if (tile < ntiles) issue_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // tile 0 (this wave's share) and the weights
  int buf = 0;
  for (; tile < ntiles; tile += (int)gridDim.x, buf ^= 1) {
    // every wave confirmed its share of this tile before its previous stores (below) ... now everybody's is in
    __builtin_amdgcn_s_barrier();
    // every wave has finished reading the other buffer (tile - grid): refill it with the next tile
    if (tile + (int)gridDim.x < ntiles) issue_tile(tile + (int)gridDim.x, buf ^ 1);

    const int n = fd_div(tile, fd_txy);
    const int rem = tile - n * (tiles_x * tiles_y);
    const int ty = fd_div(rem, fd_tx);
    const int tx = rem - ty * tiles_x;
    const int oy0 = ty * kS3T, ox0 = tx * kS3T;
    const char* Xb = Xl + buf * XBytes;

    for (int wv = wave; wv < 8; wv += NW) {          // "wave row" wv = output rows 2 wv, 2 wv + 1 of the tile
    const int oyl = 2 * wv + (r >> 4), oxl = r & 15;
    const int q0 = oyl * kS3I + oxl;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // K step = 16 channels of one tap: chunk pair (2 kk, 2 kk + 1) of the pixel, lane half h takes chunk 2 kk + h
    V8 fa[2][CINH], fb[2][CINH][NT];
#define YV4_S3_LOAD(TAP, SET)                                                                          \
  {                                                                                                    \
    const int dy_ = (TAP) / 3, dx_ = (TAP) - dy_ * 3;                                                  \
    const int q_ = q0 + dy_ * kS3I + dx_;                                                              \
    const int sw_ = xswz(q_);                                                                          \
    const char* sp_ = Xb + q_ * PixB;                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < CINH; ++kk) {                                              \
      fa[SET][kk] = *reinterpret_cast<const V8*>(sp_ + (((kk * 2 + h) ^ sw_) << 4));                   \
      _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                   \
          fb[SET][kk][t] = *reinterpret_cast<const V8*>(                                               \
              Wl + wrow[t] + ((((unsigned)((TAP) * Cpp + kk * 2 + h)) ^ whs[t]) << 4));                \
    }                                                                                                  \
  }
    YV4_S3_LOAD(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) {
        if (tap & 1) { YV4_S3_LOAD(tap + 1, 0); } else { YV4_S3_LOAD(tap + 1, 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < CINH; ++kk)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = Elem<BF16>::mfma(fa[tap & 1][kk], fb[tap & 1][kk][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef YV4_S3_LOAD

    // ---- epilogue: lane (r, h) holds channel 32 t + r of this wave's pixels m = (e & 3) + 8 (e >> 2) + 4 h.  After the
    // pair exchange the even lane of a channel pair owns output row 2w, the odd lane row 2w + 1, columns
    // (j & 3) + 8 (j >> 2) + 4 h, two channels per dword -- for the residual loads and for the stores.
    const int orow = 2 * wv + (odd ? 1 : 0);
    const bool row_ok = oy0 + orow < p.Ho;
    const bool full_x = ox0 + kS3T <= p.Wo;
    const size_t pix0 = (size_t)(n * p.Ho + oy0 + orow) * p.Wo + ox0 + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = t * 32 + r;
      if (c >= p.Cout) continue;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = acc[t][e] * s1[t] + t1[t];
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          v[e] = (float)(T)v[e];
          const int m = (e & 3) + 8 * (e >> 2) + 4 * h;       // this wave's pixel: row 2w + (m >> 4), column m & 15
          const bool in = oy0 + 2 * wv + (m >> 4) < p.Ho && ox0 + (m & 15) < p.Wo;
          st_su[t] += in ? v[e] : 0.f;
          st_sq[t] += in ? v[e] * v[e] : 0.f;
        }
      }
      switch (p.act1) {
        case YV4_ACT_MISH: mish_fast_row(v); break;
        case YV4_ACT_LEAKY:
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope1;
          break;
        case YV4_ACT_SWISH:
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = apply_act(v[e], YV4_ACT_SWISH, 0.f);
          break;
        default: break;
      }
      if (p.res) {
        // dword (two channels) per lane and column of ITS row; the partner lane's dwords supply the other row
        const T* rb = reinterpret_cast<const T*>(p.res) + (pix0 * p.r_cs + p.r_co + (c & ~1));
        unsigned mine[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = (j & 3) + 8 * (j >> 2);
          mine[j] = (row_ok && (full_x || ox0 + col + 4 * h < p.Wo)) ? *reinterpret_cast<const unsigned*>(rb + (size_t)col * p.r_cs) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned theirs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine[j], 0xB1, 0xF, 0xF, false);
          // even lane (channel c): row 0 value = low half of mine, row 1 value = low half of theirs;
          // odd lane (channel c = pair's high half): row 1 value = high half of mine, row 0 value = high half of theirs
          const unsigned a = odd ? (theirs >> 16) : (mine[j] & 0xffffu);
          const unsigned b = odd ? (mine[j] >> 16) : (theirs & 0xffffu);
          unsigned short ua = (unsigned short)a, ub = (unsigned short)b;
          v[j] += (float)__builtin_bit_cast(T, ua);
          v[j + 8] += (float)__builtin_bit_cast(T, ub);
        }
      }
      if (has2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * s2[t] + t2[t];
        switch (p.act2) {
          case YV4_ACT_MISH: mish_fast_row(v); break;
          case YV4_ACT_LEAKY:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope2;
            break;
          case YV4_ACT_SWISH:
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = apply_act(v[e], YV4_ACT_SWISH, 0.f);
            break;
          default: break;
        }
      }
      unsigned pk[8];
      pair_pack16<T>(v, odd, pk);
      // the next tile's DMA has had the whole tile to land: confirm it BEFORE the stores join vmcnt (a wait after
      // them would sit out their acknowledgement once per tile)
      if (t == 0 && wv == wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      T* yb = reinterpret_cast<T*>(p.y) + (pix0 * p.y_cs + p.y_co + (c & ~1));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = (j & 3) + 8 * (j >> 2);
        if (row_ok && (full_x || ox0 + col + 4 * h < p.Wo)) *reinterpret_cast<unsigned*>(yb + (size_t)col * p.y_cs) = pk[j];
      }
    }
    }   // wave rows
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.stats) {
    const StatRep rep = stat_rep(p.stats, (unsigned)((blockIdx.x * NW + wave)), p.Cout);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float su = st_su[t], sq = st_sq[t];
      su += __shfl_xor(su, 32);
      sq += __shfl_xor(sq, 32);
      const int c = t * 32 + r;
      if (h == 0 && c < p.Cout) {
        stat_add(rep, c, su);
        stat_add(rep, p.Cout + c, sq);
      }
    }
  }
}

static size_t s3_lds_bytes(int Cin, int Cout) {
  const size_t xb = ((size_t)kS3Pix * Cin * 2 + 1023) & ~(size_t)1023;
  return 2 * xb + (size_t)((Cout + 31) / 32 * 32) * 9 * Cin * 2;
}

// Is this layer in the kernel's domain?
bool conv3x3_small_applies(const ConvArgsH& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && !a.ys_on && !a.out_f32 &&
         (a.Cin == 16 || a.Cin == 32 || a.Cin == 64) && a.Kw == 9 * a.Cin && a.Cout >= 16 &&
         a.Cout <= 64 && (a.Cout & 1) == 0 && ((a.y_cs | a.y_co) & 1) == 0 &&
         (a.res == nullptr || ((a.r_cs | a.r_co) & 1) == 0) && s3_lds_bytes(a.Cin, a.Cout) <= 160 * 1024;
}

template <bool BF16, int CINH, int NT, int NW>
static int launch_s3_nw(const ConvArgsH& a, hipStream_t stream) {
  const int tiles_x = (a.Wo + kS3T - 1) / kS3T, tiles_y = (a.Ho + kS3T - 1) / kS3T;
  const long long nt = (long long)a.N * tiles_x * tiles_y;
  if (nt >= (1LL << 31)) {
    set_error("conv3x3 small: too many tiles");
    return YV4_E_UNSUPPORTED;
  }
  const size_t lds = s3_lds_bytes(CINH * 16, NT * 32);
  const long long xb = (long long)a.N * a.H * a.W * a.x_cs * 2;
  auto kern = conv3x3_small_kernel<BF16, CINH, NT, NW>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), lds, "conv3x3_small_h16")) return rc;
  // workgroups per CU: by LDS (160 KB) and, for the 4-wave form, at most 4 (16 waves, <= 128 VGPRs each)
  int per_cu = 1;
  if (NW == 4) {
    per_cu = (int)(160 * 1024 / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
  }
  const long long slots = 256LL * per_cu;
  const int grid = nt < slots ? (int)nt : (int)slots;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a, (unsigned)xb, tiles_x, tiles_y, (int)nt,
                     make_fastdiv((unsigned)tiles_x), make_fastdiv((unsigned)(tiles_x * tiles_y)));
  YV4_CHECK_LAUNCH("conv3x3_small_h16");
  return YV4_OK;
}

// 4-wave workgroups when at least two fit a CU (the 16- and 32-channel layers); YV4_S3_WAVES=8 keeps one of 8
template <bool BF16, int CINH, int NT>
static int launch_s3(const ConvArgsH& a, hipStream_t stream) {
  static const bool allow4 = YV4_ENV_INT("YV4_S3_WAVES", 4) != 8;
  if (allow4 && 2 * s3_lds_bytes(CINH * 16, NT * 32) <= 160 * 1024) return launch_s3_nw<BF16, CINH, NT, 4>(a, stream);
  return launch_s3_nw<BF16, CINH, NT, 8>(a, stream);
}

int conv3x3_small_launch(const ConvArgsH& a, bool bf16, hipStream_t s) {
  const int nt = a.Cout > 32 ? 2 : 1;
#define YV4_S3_CASE(CH)                                                                     \
  if (nt == 2) return bf16 ? launch_s3<true, CH, 2>(a, s) : launch_s3<false, CH, 2>(a, s);  \
  return bf16 ? launch_s3<true, CH, 1>(a, s) : launch_s3<false, CH, 1>(a, s);
  if (a.Cin == 16) { YV4_S3_CASE(1) }
  if (a.Cin == 32) { YV4_S3_CASE(2) }
  YV4_S3_CASE(4)
#undef YV4_S3_CASE
}

}  // namespace yv4
