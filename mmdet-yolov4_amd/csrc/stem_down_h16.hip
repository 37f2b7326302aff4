// The first two layers of CSPDarknet as ONE kernel for the 16-bit inference plans (gfx950):
//   stem  Conv(3 -> C1, 3x3, stride 1) + BN + act      (darknetcsp.py:357-366 `conv` stage, Conv = :15-35)
//   down  Conv(C1 -> C2, 3x3, stride 2) + BN + act     (the next stage's `conv_downscale`, darknetcsp.py:290-300)
// NCHW fp32 image in, NHWC 16-bit activations (N, H/2, W/2, C2) out.
//
// Why.  These two launches (plus the NCHW -> NHWC4 repack feeding them) are the most HBM-bound of the network: the
// stem writes H*W*C1 16-bit values that the stride-2 conv immediately reads back -- 1.5 GB of the 2.4 GB the three
// launches move at YOLOv4-L 608 x 608 x 32 (2.9 of 3.6 GB at YOLOv4-s 416 x 416 x 256) -- and the fp32-MFMA stem
// kernel (18 x 64-cycle MFMAs per 32 pixels for 27 useful multiplies each) is matrix-pipe-bound on top of that.
// Fused, the stem's output never leaves the CU.
//
// One persistent 8-wave workgroup per CU walks tiles of 15 rows x 16 columns of the OUTPUT map (the stem region of
// such a tile is 31 x 33 = 1023 pixels = 32 MFMA row tiles, four per wave; a 16 x 16 tile needs 1089 = 34.03, i.e.
// five rounds for eight waves).  Per tile:
//   A. the 33 x 35 x 3 input patch (NCHW planes, zero outside the image) is split into two 16-bit terms
//      x = hi + lo and written to LDS as 4-channel pixels (the loads of the NEXT tile's patch are issued here and
//      land during B and C);
//   B. the stem on the 31 x 33 pixels the tile needs, as 16-bit MFMAs with a three-term split of the fp32 product
//      (hi*Whi + lo*Whi + hi*Wlo, K = 27 (term, tap) slots x 4 channels = 7 steps of 32x32x16; the dropped lo*Wlo term
//      is 2^-16 of the product for bf16, 2^-22 for fp16, i.e. the result is the fp32 convolution of the fp32 stem
//      kernel to within fp32 accumulation noise).  Weights as the MFMA's A operand stay in registers for the whole
//      kernel; a lane ends with 16 channels of ONE pixel, applies BN + act, ZEROES pixels outside the image (the
//      stride-2 conv pads its input, not the image) and writes them to the LDS stem tile, even and odd columns in
//      separate planes so that the stride-2 reads of C are unit-stride, 16-byte chunks XOR-swizzled against bank
//      conflicts;
//   C. the stride-2 conv from LDS (A operand: stem tile, B operand: its weights, resident in LDS since the start of
//      the kernel), BN + act, dword stores of channel pairs (pair_pack16), 64-byte segments per pixel.
// Two workgroup barriers per tile.
#include "conv_h16_common.h"

namespace yv4 {

// 8 waves per workgroup; the narrow variant (C1 = 16) runs two 4-wave workgroups per CU (template parameter NW)
constexpr int kSdTx = 16;                       // output tile columns; rows: template parameter TY (15 or 16)
constexpr int kSdSC = 2 * kSdTx + 1;            // stem tile columns (33)
constexpr int kSdPC = kSdSC + 2;                // input patch columns (35)
constexpr int kSdPPitch = 36;                   // patch row pitch in pixels
constexpr int kSdSCols = kSdTx + 1;             // columns per parity plane (17)
constexpr int sd_plane_bytes(int ty) { return (2 * ty + 3) * kSdPPitch * 8; }   // one term's patch (4 x 16-bit per pixel)
constexpr int sd_stem_pixels(int ty) { return 2 * (2 * ty + 1) * kSdSCols; }    // stem tile pixels in LDS

struct StemDownArgs {
  const float* x;       // (N, 3, H, W) fp32
  const float* w1;      // (C1, 36) fp32: [cout][tap * 4 + ci], ci = 3 zero
  const float* s1; const float* t1;
  const void* w2;       // (C2, 9 * C1) 16-bit: [cout][(tap, ci)]
  const float* s2; const float* t2;
  void* y;
  int N, H, W, Ho, Wo, C1, C2, y_cs, y_co;
  int act1, act2; float slope1, slope2;
  int tiles_x, tiles_y, ntiles;
  FastDiv fd_tx, fd_ty;
  int ablate;   // measurement only (YV4_SD_ABLATE): 1 no phase B, 2 no phase C, 4 no activation, 8 no stores, 16 no patch loads
};

__device__ __forceinline__ void sd_act16(float (&v)[16], int act, float slope) {
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

// C1H = C1 / 16 (1 or 2), NT2 = C2 / 32 (1 or 2), TY = output rows per tile, NW = waves per workgroup (8 or 4).
// NW = 4: half the waves per workgroup and (LDS permitting: C1 = 16 needs 66 KB) two workgroups per CU, which drift out
// of phase -- one in its stem phase while the other convolves -- instead of eight waves meeting at every barrier.
template <bool BF16, int C1H, int NT2, int TY, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void stem_down_kernel(StemDownArgs p) {
  constexpr int kSdThreads = NW * 64;
  constexpr int kPasses = (( (2 * TY + 3) * kSdPC) + kSdThreads - 1) / kSdThreads;      // phase A passes over the patch
  constexpr int kSdTy = TY;
  constexpr int kSdSR = 2 * TY + 1;               // stem tile rows
  constexpr int kSdPR = kSdSR + 2;                // input patch rows
  constexpr int kSdPPlane = sd_plane_bytes(TY);
  constexpr int kSdSPix = sd_stem_pixels(TY);
  constexpr int kRowTiles = (kSdSR * kSdSC + 31) / 32;
  static_assert(kSdPR * kSdPC <= kPasses * kSdThreads, "phase A covers the patch");
  typedef typename Elem<BF16>::T T;
  typedef typename Elem<BF16>::V8 V8;
  typedef T T4 __attribute__((ext_vector_type(4)));
  constexpr int C1 = C1H * 16;
  constexpr int C2 = NT2 * 32;
  constexpr int K2 = 9 * C1;
  constexpr int W2Pitch = K2 * 2;           // bytes per weight row
  constexpr int W2Cpr = K2 / 8;             // 16-byte chunks per weight row (36 / 18)
  constexpr int SPix = C1 * 2;              // bytes per stem pixel
  extern __shared__ __attribute__((aligned(16))) char smem_sd[];
  char* Pl = smem_sd;                                   // [2 terms][33][36] x 8 B
  char* Sl = smem_sd + 2 * kSdPPlane;                   // [2 parities][31][17] x SPix, chunks swizzled
  char* Wl = Sl + ((kSdSPix * SPix + 15) & ~15);        // [C2][K2] 16-bit, chunks swizzled
  char* Al = Wl + C2 * W2Pitch;                         // stem affine, 64 floats

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;
  const int h = lane >> 5;

  // ---- once: the stride-2 conv's weights into LDS
  for (int c = tid; c < C2 * W2Cpr; c += kSdThreads) {
    const int row = c / W2Cpr, kc = c - row * W2Cpr;
    const int swz = C1H == 2 ? ((row >> 2) & 3) : ((row >> 3) & 1);
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w2) + (size_t)row * W2Pitch + kc * 16);
    *reinterpret_cast<uint4*>(Wl + row * W2Pitch + ((kc ^ swz) << 4)) = v;
  }
  // ---- once: the stem's weights as A fragments (row r = output channel, 8 K values per lane and step).
  // K slot = term * 9 + tap (27 slots + one empty), 4 channels per slot; step s covers slots 4s .. 4s+3, lane half h
  // the slots 4s + 2h and 4s + 2h + 1.  term 0: Whi (against x hi), 1: Whi (against x lo), 2: Wlo (against x hi).
  V8 wfrag[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int slot = 4 * s + 2 * h + u;
      const int term = slot / 9, tap = slot - term * 9;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        float w = 0.f;
        if (slot < 27 && r < C1 && ci < 3) w = p.w1[r * 36 + tap * 4 + ci];
        const T hi = (T)w;
        const T lo = (T)(w - (float)hi);
        wfrag[s][u * 4 + ci] = term == 2 ? lo : hi;
      }
    }
  }
  // per-lane patch offsets of the two slots of every step (x term plane + tap displacement); slot 27 reads anything
  // (its weights are zero) -- but not NaN garbage: it is masked below
  int poff[7][2];
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int slot = 4 * s + 2 * h + u;
      const int term = slot / 9, tap = slot - term * 9;
      const int dy = tap / 3, dx = tap - dy * 3;
      const int xplane = term == 1 ? 1 : 0;
      poff[s][u] = slot < 27 ? xplane * kSdPPlane + (dy * kSdPPitch + dx) * 8 : 0;
    }
  const bool last_slot_empty = h == 1;     // step 6, u = 1 is slot 27
  // stem affine: [scale (32) | shift (32)] floats in LDS, read back four channels at a time in the epilogue (as
  // registers the 32 values cost the staging room the MFMA operands need)
  if (tid < 32) {
    reinterpret_cast<float*>(Al)[tid] = tid < C1 ? p.s1[tid] : 0.f;
    reinterpret_cast<float*>(Al)[32 + tid] = tid < C1 ? p.t1[tid] : 0.f;
  }
  // down conv affine: lane = output channel 32 t + r
  float sb[NT2], tb[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t) { sb[t] = p.s2[t * 32 + r]; tb[t] = p.t2[t * 32 + r]; }

  // phase A thread map: patch pixels tid, tid + threads, ...
  int a_py[kPasses], a_px[kPasses];
#pragma unroll
  for (int k = 0; k < kPasses; ++k) {
    const int q = tid + k * kSdThreads;
    a_py[k] = q / kSdPC;
    a_px[k] = q - a_py[k] * kSdPC;
  }
  const size_t plane = (size_t)p.H * p.W;

  float pre[kPasses][3];
  auto load_patch = [&](int tile) {
    const int n = fd_div(tile, p.fd_ty);                   // tile / (tiles_x * tiles_y)
    const int rem = tile - n * (p.tiles_x * p.tiles_y);
    const int ty = fd_div(rem, p.fd_tx);
    const int tx = rem - ty * p.tiles_x;
    const int iy0 = 2 * ty * kSdTy - 2, ix0 = 2 * tx * kSdTx - 2;
    const float* xb = p.x + (size_t)n * 3 * plane;
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
      const int iy = iy0 + a_py[k], ix = ix0 + a_px[k];
      const bool ok = tid + k * kSdThreads < kSdPR * kSdPC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const size_t o = ok ? (size_t)iy * p.W + ix : 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) pre[k][c] = ok ? xb[c * plane + o] : 0.f;
    }
  };

  int tile = (int)blockIdx.x;
  if (tile < p.ntiles) load_patch(tile);
  __syncthreads();      // Wl complete

  for (; tile < p.ntiles; tile += (int)gridDim.x) {
    const int n = fd_div(tile, p.fd_ty);
    const int rem = tile - n * (p.tiles_x * p.tiles_y);
    const int ty = fd_div(rem, p.fd_tx);
    const int tx = rem - ty * p.tiles_x;
    const int oy0 = ty * kSdTy, ox0 = tx * kSdTx;

    // ---- A: split and store the patch; prefetch the next one
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
      if (tid + k * kSdThreads < kSdPR * kSdPC) {
        T4 hi, lo;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          hi[c] = (T)pre[k][c];
          lo[c] = (T)(pre[k][c] - (float)hi[c]);
        }
        hi[3] = (T)0.f; lo[3] = (T)0.f;
        const int o = (a_py[k] * kSdPPitch + a_px[k]) * 8;
        *reinterpret_cast<T4*>(Pl + o) = hi;
        *reinterpret_cast<T4*>(Pl + kSdPPlane + o) = lo;
      }
    }
    if (tile + (int)gridDim.x < p.ntiles && !YV4_ABLATE(p.ablate, 16)) load_patch(tile + (int)gridDim.x);
    __syncthreads();

    // ---- B: the stem on 31 x 33 pixels, 32 per row tile, four row tiles per wave
    for (int rt = wave; rt < kRowTiles && !YV4_ABLATE(p.ablate, 1); rt += NW) {
      const int pix = rt * 32 + r;
      const int sy = pix / kSdSC, sx = pix - sy * kSdSC;
      const char* pb = Pl + (sy * kSdPPitch + sx) * 8;
      // all 14 operand reads first, then the 7 MFMAs (left to itself the compiler reads each step's operands right
      // before its MFMA: seven exposed LDS latencies per row tile, 2x the time of the whole phase)
      T4 xa[7], xb[7];
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        xa[s] = *reinterpret_cast<const T4*>(pb + poff[s][0]);
        xb[s] = *reinterpret_cast<const T4*>(pb + poff[s][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (last_slot_empty) { xb[6][0] = (T)0.f; xb[6][1] = (T)0.f; xb[6][2] = (T)0.f; xb[6][3] = (T)0.f; }
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        V8 xf;
        xf[0] = xa[s][0]; xf[1] = xa[s][1]; xf[2] = xa[s][2]; xf[3] = xa[s][3];
        xf[4] = xb[s][0]; xf[5] = xb[s][1]; xf[6] = xb[s][2]; xf[7] = xb[s][3];
        acc = Elem<BF16>::mfma(wfrag[s], xf, acc);     // rows = channels, columns = pixels
      }
      if (pix < kSdSR * kSdSC) {
        const int ay = 2 * oy0 - 1 + sy, ax = 2 * ox0 - 1 + sx;
        const bool inside = (unsigned)ay < (unsigned)p.H && (unsigned)ax < (unsigned)p.W;
        const int q = ((sx & 1) * kSdSR + sy) * kSdSCols + (sx >> 1);
        const int swz = C1H == 2 ? ((q >> 2) & 3) : ((q >> 3) & 1);
        char* sp = Sl + q * SPix;
        if (inside) {
          // only the 8 C1H accumulator registers that hold real channels (C1 = 16: MFMA rows 16-31 are padding)
          float v[8 * C1H];
#pragma unroll
          for (int g = 0; g < 2 * C1H; ++g) {         // channels 8g + 4h .. +3
            const float4 sc = *reinterpret_cast<const float4*>(Al + (8 * g + 4 * h) * 4);
            const float4 sh = *reinterpret_cast<const float4*>(Al + (32 + 8 * g + 4 * h) * 4);
            v[4 * g + 0] = acc[4 * g + 0] * sc.x + sh.x;
            v[4 * g + 1] = acc[4 * g + 1] * sc.y + sh.y;
            v[4 * g + 2] = acc[4 * g + 2] * sc.z + sh.z;
            v[4 * g + 3] = acc[4 * g + 3] * sc.w + sh.w;
          }
          if (!YV4_ABLATE(p.ablate, 4)) {
            switch (p.act1) {
              case YV4_ACT_MISH: mish_fast_row(v); break;
              case YV4_ACT_LEAKY:
#pragma unroll
                for (int e = 0; e < 8 * C1H; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.slope1;
                break;
              case YV4_ACT_SWISH:
#pragma unroll
                for (int e = 0; e < 8 * C1H; ++e) v[e] = apply_act(v[e], YV4_ACT_SWISH, 0.f);
                break;
              default: break;
            }
          }
#pragma unroll
          for (int g = 0; g < 2 * C1H; ++g) {         // chunk g, half h
            T4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = (T)v[4 * g + u];
            *reinterpret_cast<T4*>(sp + ((g ^ swz) << 4) + h * 8) = o;
          }
        } else {                                       // outside the image: the stride-2 conv's zero padding
          T4 o;
          o[0] = (T)0.f; o[1] = (T)0.f; o[2] = (T)0.f; o[3] = (T)0.f;
#pragma unroll
          for (int g = 0; g < 2 * C1H; ++g) *reinterpret_cast<T4*>(sp + ((g ^ swz) << 4) + h * 8) = o;
        }
      }
    }
    __syncthreads();

    // ---- C: the stride-2 conv on the tile; wave w owns output rows 2w, 2w + 1 (32 pixels; row 15 is not part of
    // the tile: computed on whatever the LDS holds and never stored)
    if (!YV4_ABLATE(p.ablate, 2))
    for (int wv = wave; wv < 8; wv += NW) {          // "wave row" wv = output rows 2 wv, 2 wv + 1
      const int oyl = 2 * wv + (r >> 4), oxl = r & 15;
      const int q0 = (2 * oyl) * kSdSCols + oxl;
      f32x16 acc[NT2];
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
      // operand reads one tap ahead of the MFMAs that use them (two register sets)
      V8 fa[2][C1H], fb[2][C1H][NT2];
      unsigned wrow[NT2], whs[NT2];
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const int row = t * 32 + r;
        const int wswz = C1H == 2 ? ((row >> 2) & 3) : ((row >> 3) & 1);
        wrow[t] = (unsigned)(row * W2Pitch);
        whs[t] = (unsigned)(h ^ wswz);              // chunk kc = tap * (C1 / 8) + 2 kk + h, stored at kc ^ wswz
      }
#define YV4_SD_LOAD(TAP, SET)                                                                          \
  {                                                                                                    \
    const int dy_ = (TAP) / 3, dx_ = (TAP) - dy_ * 3;                                                  \
    const int q_ = q0 + ((dx_ & 1) * kSdSR + dy_) * kSdSCols + (dx_ >> 1);                             \
    const int swz_ = C1H == 2 ? ((q_ >> 2) & 3) : ((q_ >> 3) & 1);                                     \
    const char* sp_ = Sl + q_ * SPix;                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < C1H; ++kk) {                                               \
      fa[SET][kk] = *reinterpret_cast<const V8*>(sp_ + (((kk * 2 + h) ^ swz_) << 4));                  \
      _Pragma("unroll") for (int t = 0; t < NT2; ++t)                                                  \
          fb[SET][kk][t] = *reinterpret_cast<const V8*>(Wl + wrow[t] + ((((unsigned)((TAP) * (C1 / 8) + kk * 2)) ^ whs[t]) << 4)); \
    }                                                                                                  \
  }
      YV4_SD_LOAD(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
          if (tap & 1) { YV4_SD_LOAD(tap + 1, 0); } else { YV4_SD_LOAD(tap + 1, 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < C1H; ++kk)
#pragma unroll
          for (int t = 0; t < NT2; ++t) acc[t] = Elem<BF16>::mfma(fa[tap & 1][kk], fb[tap & 1][kk][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef YV4_SD_LOAD
      // epilogue: lane (r, h) holds channel 32 t + r of this wave's pixels m = (e & 3) + 8 (e >> 2) + 4 h.  After the pair
      // exchange the even lane of a channel pair stores output row 2w, the odd lane row 2w + 1, columns
      // (j & 3) + 8 (j >> 2) + 4 h, as dwords (two channels).
      const bool odd = r & 1;
      const int orow = 2 * wv + (odd ? 1 : 0);
      const bool row_ok = orow < kSdTy && oy0 + orow < p.Ho;
      const bool full_x = ox0 + kSdTx <= p.Wo;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = acc[t][e] * sb[t] + tb[t];
        if (!YV4_ABLATE(p.ablate, 4)) sd_act16(v, p.act2, p.slope2);
        unsigned pk[8];
        pair_pack16<T>(v, odd, pk);
        T* yb = reinterpret_cast<T*>(p.y) + (((size_t)(n * p.Ho + oy0 + orow) * p.Wo + ox0 + 4 * h) * p.y_cs + p.y_co + t * 32 + (r & ~1));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = (j & 3) + 8 * (j >> 2);
          if (row_ok && (full_x || ox0 + col + 4 * h < p.Wo) && (!YV4_ABLATE(p.ablate, 8) || pk[j] == 0x12345678u))
            *reinterpret_cast<unsigned*>(yb + col * p.y_cs) = pk[j];
        }
      }
    }
    // (the next tile's phase A writes the patch, last read in B; its phase B writes the stem tile only after the
    // barrier that ends A, which every wave reaches after its C)
  }
}

template <bool BF16, int C1H, int NT2, int TY, int NW>
static int launch_sd_ty(const StemDownArgs& a, hipStream_t stream) {
  constexpr int C1 = C1H * 16, C2 = NT2 * 32;
  const size_t lds = 2 * (size_t)sd_plane_bytes(TY) + (((size_t)sd_stem_pixels(TY) * C1 * 2 + 15) & ~(size_t)15) +
                     (size_t)C2 * 9 * C1 * 2 + 256;
  auto kern = stem_down_kernel<BF16, C1H, NT2, TY, NW>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), lds, "stem_down_h16")) return rc;
  const int slots = NW == 4 ? 512 : 256;
  const int grid = a.ntiles < slots ? a.ntiles : slots;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  YV4_CHECK_LAUNCH("stem_down_h16");
  return YV4_OK;
}

// rows per tile: 16 (default) or 15 (YV4_SD_TY=15: 32 instead of 35 stem row tiles per tile, but 7 % more tiles)
static const int g_sd_ty = YV4_ENV_INT("YV4_SD_TY", 16) == 15 ? 15 : 16;
template <bool BF16, int C1H, int NT2>
static int launch_sd(StemDownArgs a, hipStream_t stream) {
  a.tiles_y = (a.Ho + g_sd_ty - 1) / g_sd_ty;
  a.ntiles = a.N * a.tiles_x * a.tiles_y;
  a.fd_ty = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
  // YV4_SD_WAVES=8 keeps one 8-wave workgroup per CU for the narrow variant too (A/B measurement)
  static const bool narrow4 = YV4_ENV_INT("YV4_SD_WAVES", 4) != 8;
  if (C1H == 1 && narrow4 && g_sd_ty == 16) return launch_sd_ty<BF16, C1H, NT2, 16, 4>(a, stream);
  return g_sd_ty == 15 ? launch_sd_ty<BF16, C1H, NT2, 15, 8>(a, stream) : launch_sd_ty<BF16, C1H, NT2, 16, 8>(a, stream);
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_stem_down_fwd_h16(int dtype, const float* x_nchw, int N, int H, int W, const float* w1,
                                     const float* scale1, const float* shift1, int C1, int act1, float slope1,
                                     const void* w2, const float* scale2, const float* shift2, int C2, int act2,
                                     float slope2, void* y, int y_cstride, int y_coff, void* stream) {
  YV4_REQUIRE(x_nchw && w1 && scale1 && shift1 && w2 && scale2 && shift2 && y, "stem_down: null argument");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "stem_down: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(N > 0 && H > 0 && W > 0, "stem_down: empty shape");
  YV4_REQUIRE((C1 == 16 || C1 == 32) && (C2 == 32 || C2 == 64), "stem_down: built for C1 in {16, 32}, C2 in {32, 64} (got %d, %d)",
              C1, C2);
  YV4_REQUIRE(act1 >= 0 && act1 <= YV4_ACT_SWISH && act2 >= 0 && act2 <= YV4_ACT_SWISH, "stem_down: unknown activation id");
  YV4_REQUIRE(y_coff >= 0 && y_coff + C2 <= y_cstride, "stem_down: output view exceeds its pixel stride");
  YV4_REQUIRE(((y_cstride | y_coff) & 1) == 0 && ((uintptr_t)y & 3) == 0, "stem_down: the output view must be dword-aligned");
  YV4_REQUIRE(((uintptr_t)w2 & 15) == 0, "stem_down: w2 must be 16-byte aligned");
  StemDownArgs a;
  a.x = x_nchw; a.w1 = w1; a.s1 = scale1; a.t1 = shift1; a.w2 = w2; a.s2 = scale2; a.t2 = shift2; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.C1 = C1; a.C2 = C2;
  a.y_cs = y_cstride; a.y_co = y_coff; a.act1 = act1; a.act2 = act2; a.slope1 = slope1; a.slope2 = slope2;
  a.tiles_x = (a.Wo + kSdTx - 1) / kSdTx; a.tiles_y = (a.Ho + 14) / 15;     // (tiles_y: upper bound, set by launch_sd)
  const long long nt = (long long)N * a.tiles_x * a.tiles_y;
  YV4_REQUIRE(nt < (1LL << 31) && (long long)N * a.Ho * a.Wo < (1LL << 31), "stem_down: too many tiles / pixels");
  a.ntiles = (int)nt;
  static const int ablate = YV4_ENV_INT("YV4_SD_ABLATE", 0);
  a.ablate = ablate;
  a.fd_tx = make_fastdiv((unsigned)a.tiles_x);
  a.fd_ty = make_fastdiv((unsigned)(a.tiles_x * a.tiles_y));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool bf = dtype == YV4_BF16;
  if (C1 == 32 && C2 == 64) return bf ? launch_sd<true, 2, 2>(a, s) : launch_sd<false, 2, 2>(a, s);
  if (C1 == 32 && C2 == 32) return bf ? launch_sd<true, 2, 1>(a, s) : launch_sd<false, 2, 1>(a, s);
  if (C1 == 16 && C2 == 64) return bf ? launch_sd<true, 1, 2>(a, s) : launch_sd<false, 1, 2>(a, s);
  return bf ? launch_sd<true, 1, 1>(a, s) : launch_sd<false, 1, 1>(a, s);
}
