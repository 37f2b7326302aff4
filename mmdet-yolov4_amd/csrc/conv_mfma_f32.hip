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
#include "yv4_common.h"

namespace yv4 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;   // K slice staged per step (floats)
constexpr int kLDK = 36;  // LDS row pitch in floats: 32 + 4 pad (144 B, 16B aligned)
constexpr int kThreads = 256;

struct ConvArgs {
  const float* x;
  const float* w;
  const float* s1;
  const float* t1;
  const float* s2;
  const float* t2;
  const float* res;
  float* y;
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int x_cs, x_co, y_cs, y_co, r_cs, r_co;
  int act1, act2;
  float slope1, slope2;
  int M, K, Kw;  // Kw: row pitch of w (== K)
  int tiles_n;
};

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

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_conv(const ConvArgs& a, bool uniform_tap, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * kLDK * sizeof(float);
  ConvArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const long long tiles = (long long)tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) {
    set_error("conv: grid of %lld tiles out of range", tiles);
    return YV4_E_INVALID;
  }
  auto kern_u = conv_mfma_f32_kernel<BM, BN, WAVES_M, WAVES_N, true>;
  auto kern_g = conv_mfma_f32_kernel<BM, BN, WAVES_M, WAVES_N, false>;
  static bool attr_done = false;  // benign race: idempotent
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern_u),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern_g),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  if (uniform_tap)
    hipLaunchKernelGGL(kern_u, dim3((unsigned)tiles), dim3(kThreads), lds, stream, p);
  else
    hipLaunchKernelGGL(kern_g, dim3((unsigned)tiles), dim3(kThreads), lds, stream, p);
  YV4_CHECK_LAUNCH("conv_mfma_f32");
  return YV4_OK;
}

static int pick_tile(long long M, int Cout) {
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

extern "C" int yv4_conv_pick_tile(const yv4_conv_desc* d) {
  if (!d) return YV4_TILE_AUTO;
  return pick_tile((long long)d->N * d->Ho * d->Wo, d->Cout);
}

extern "C" int yv4_conv_bn_act_fwd(const yv4_conv_desc* d, const float* x, const float* w,
                                   const float* scale1, const float* shift1,
                                   const float* scale2, const float* shift2,
                                   const float* residual, float* y, void* stream) {
  YV4_REQUIRE(d && x && w && scale1 && shift1 && y, "conv: null argument");
  YV4_REQUIRE((scale2 == nullptr) == (shift2 == nullptr), "conv: scale2/shift2 must come together");
  YV4_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: empty shape");
  YV4_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel/stride/pad");
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

  const bool uniform = (d->Cin % kBK) == 0;
  int tile = d->tile == YV4_TILE_AUTO ? pick_tile(M, d->Cout) : d->tile;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (tile) {
    case YV4_TILE_128x128: return launch_conv<128, 128, 2, 2>(a, uniform, s);
    case YV4_TILE_128x64: return launch_conv<128, 64, 2, 2>(a, uniform, s);
    case YV4_TILE_64x128: return launch_conv<64, 128, 2, 2>(a, uniform, s);
    case YV4_TILE_64x64: return launch_conv<64, 64, 2, 2>(a, uniform, s);
    default:
      set_error("conv: unknown tile id %d", tile);
      return YV4_E_INVALID;
  }
}
