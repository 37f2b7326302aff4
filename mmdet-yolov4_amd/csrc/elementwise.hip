// HBM-bound helpers of the YOLOv4 path on gfx950: standalone Mish fwd/bwd, the
// NCHW<->NHWC adaptors, the SPP max-pools and the nearest-resample-into-concat copy.
// All of them move 16 bytes per lane and are sized to fill 256 CUs.
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

#include "yv4_common.h"
#include "spp_lds.h"

namespace yv4 {

// ---------------------------------------------------------------------------------
// Mish forward / backward.  Reference: mmdet/ops/mish_cuda/src/mish.h:16-29 (math),
// src/kernel/mish_cuda.cu:25-71 (grid-stride elementwise kernels; half/bf16 computed
// in float, fp64 native).  The backward recomputes everything from the saved INPUT.
// ---------------------------------------------------------------------------------
template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct Cvt<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct Cvt<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};

__device__ __forceinline__ float mish_fwd_ref(float x) {
  // literal form of mish.h:17 (kept for the standalone op so that it follows the
  // reference's own arithmetic; the conv epilogue uses the algebraically equal form)
  return x * tanhf(x < 20.f ? log1pf(expf(x)) : x);
}
__device__ __forceinline__ float mish_bwd_ref(float g, float x) {
  const float sp = x < 20.f ? log1pf(expf(x)) : x;
  const float grad_sp = 1.f - expf(-sp);
  const float tsp = tanhf(sp);
  const float grad_tsp = (1.f - tsp * tsp) * grad_sp;
  return g * (x * grad_tsp + tsp);
}
__device__ __forceinline__ double mish_fwd_ref(double x) {
  return x * tanh(x < 20.0 ? log1p(exp(x)) : x);
}
__device__ __forceinline__ double mish_bwd_ref(double g, double x) {
  const double sp = x < 20.0 ? log1p(exp(x)) : x;
  const double grad_sp = 1.0 - exp(-sp);
  const double tsp = tanh(sp);
  const double grad_tsp = (1.0 - tsp * tsp) * grad_sp;
  return g * (x * grad_tsp + tsp);
}

// VEC elements of T per 16-byte access.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void mish_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, size_t n) {
  struct alignas(16) Pack { T v[VEC]; };
  const size_t nvec = n / VEC;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    Pack a = reinterpret_cast<const Pack*>(in)[i];
    Pack o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o.v[k] = Cvt<T>::st(mish_fwd_ref(Cvt<T>::ld(a.v[k])));
    reinterpret_cast<Pack*>(out)[i] = o;
  }
  // tail (n % VEC) by the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < n - nvec * VEC) {
    const size_t i = nvec * VEC + threadIdx.x;
    out[i] = Cvt<T>::st(mish_fwd_ref(Cvt<T>::ld(in[i])));
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void mish_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ in,
                                                       T* __restrict__ gin, size_t n) {
  struct alignas(16) Pack { T v[VEC]; };
  const size_t nvec = n / VEC;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    Pack g = reinterpret_cast<const Pack*>(gout)[i];
    Pack a = reinterpret_cast<const Pack*>(in)[i];
    Pack o;
#pragma unroll
    for (int k = 0; k < VEC; ++k) o.v[k] = Cvt<T>::st(mish_bwd_ref(Cvt<T>::ld(g.v[k]), Cvt<T>::ld(a.v[k])));
    reinterpret_cast<Pack*>(gin)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - nvec * VEC) {
    const size_t i = nvec * VEC + threadIdx.x;
    gin[i] = Cvt<T>::st(mish_bwd_ref(Cvt<T>::ld(gout[i]), Cvt<T>::ld(in[i])));
  }
}

__global__ __launch_bounds__(256) void mish_fwd_f64_kernel(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = mish_fwd_ref(in[i]);
}
__global__ __launch_bounds__(256) void mish_bwd_f64_kernel(const double* __restrict__ gout, const double* __restrict__ in,
                                                           double* __restrict__ gin, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) gin[i] = mish_bwd_ref(gout[i], in[i]);
}

static inline unsigned ew_grid(size_t work_items) {
  // memory-bound: cap at 256 CUs x 8 workgroups and grid-stride the rest
  size_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (unsigned)g;
}

// ---------------------------------------------------------------------------------
// NCHW -> NHWC (with channel zero-padding) through an LDS transpose tile.
// One workgroup: 64 pixels x all C channels (C small: the 3-channel image) or, for
// general C, a 64-pixel x 64-channel tile.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int dst_cs, int dst_co, int cpad) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63;  // pixel on load, channel on store
  const int ty = threadIdx.x >> 6;  // 0..3
  const float* s = src + (size_t)n * C * HW;
  for (int c = ty; c < 64; c += 4) {
    const int ch = c0 + c;
    const int px = p0 + tx;
    tile[c][tx] = (ch < C && px < HW) ? s[(size_t)ch * HW + px] : 0.f;
  }
  __syncthreads();
  float* d = dst + (size_t)n * HW * dst_cs + dst_co;
  for (int pp = ty; pp < 64; pp += 4) {
    const int px = p0 + pp;
    const int ch = c0 + tx;
    if (px < HW && ch < C + cpad) d[(size_t)px * dst_cs + ch] = tile[tx][pp];
  }
}

// Small-C form (the 3-channel image, C + pad == 4): one thread per pixel reads its C planar
// values (coalesced per plane) and writes one float4 -- no LDS tile, ~4x the rate of the generic
// kernel on the 142 MB input batch.
// grid (pixel blocks, image): no per-pixel 64-bit division (the flat-index form spent ~100 VALU instructions per
// pixel on `i / HW` and ran at 1.1 TB/s)
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                            int HW, int N, int dst_cs, int dst_co) {
  const int n = blockIdx.y;
  const float* s0 = src + (size_t)n * C * HW;
  float* d0 = dst + (size_t)n * HW * dst_cs + dst_co;
  for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) v[c] = s0[(size_t)c * HW + px];
    *reinterpret_cast<float4*>(d0 + (size_t)px * dst_cs) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int src_cs, int src_co) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63;
  const int ty = threadIdx.x >> 6;
  const float* s = src + (size_t)n * HW * src_cs + src_co;
  for (int pp = ty; pp < 64; pp += 4) {
    const int px = p0 + pp;
    const int ch = c0 + tx;
    tile[pp][tx] = (px < HW && ch < C) ? s[(size_t)px * src_cs + ch] : 0.f;
  }
  __syncthreads();
  float* d = dst + (size_t)n * C * HW;
  for (int c = ty; c < 64; c += 4) {
    const int ch = c0 + c;
    const int px = p0 + tx;
    if (ch < C && px < HW) d[(size_t)ch * HW + px] = tile[tx][c];
  }
}

// ---------------------------------------------------------------------------------
// SPP: MaxPool2d(k, stride=1, padding=k/2) for k = 5, 9, 13 of one NHWC map, written
// next to the input inside the cat buffer (darknetcsp.py:203-206,222-226).  One thread
// owns (n, y, x, 4 channels) and walks the 13x13 window once, folding each tap into
// the three nested maxima (window 5 within 9 within 13); -inf padding as in torch.
// The 19x19x512 map re-reads are L1/L2 hits; algorithmic traffic is 5*C*H*W floats.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

__global__ __launch_bounds__(256) void spp_pool_kernel(float* __restrict__ buf, int N, int H, int W, int C4,
                                                       int cs, int co, int C) {
  const size_t total = (size_t)N * H * W * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float ninf = -__builtin_huge_valf();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const float* base = buf + (size_t)n * H * W * cs + co + c4 * 4;
    float4 m5 = make_float4(ninf, ninf, ninf, ninf), m9 = m5, m13 = m5;
    for (int dy = -6; dy <= 6; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const int ady = dy < 0 ? -dy : dy;
      for (int dx = -6; dx <= 6; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const int adx = dx < 0 ? -dx : dx;
        const int rad = ady > adx ? ady : adx;
        const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * cs);
        m13 = max4(m13, v);
        if (rad <= 4) m9 = max4(m9, v);
        if (rad <= 2) m5 = max4(m5, v);
      }
    }
    float* o = buf + ((size_t)(n * H + y) * W + x) * cs + co + c4 * 4;
    *reinterpret_cast<float4*>(o + C) = m5;
    *reinterpret_cast<float4*>(o + 2 * C) = m9;
    *reinterpret_cast<float4*>(o + 3 * C) = m13;
  }
}

// MaxPool2d(5, 1, 2) of one channel slice of the cat buffer into another: MaxPool(9) = MaxPool(5) applied twice and
// MaxPool(13) three times, exactly (max is associative and idempotent, the implicit padding is -inf), so the SPP
// block is three 25-tap passes over a 24 MB slice instead of one 169-tap pass (208 -> ~70 us at batch 32).
// grid (x * C4 blocks, n * H + y): no per-thread index division beyond one 32-bit divide.
__global__ __launch_bounds__(256) void pool5_kernel(float* __restrict__ buf, int H, int W, int C4, int cs, int src_co,
                                                    int dst_co) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * C4) return;
  const int x = idx / C4, c4 = idx - x * C4;
  const int row = blockIdx.y;            // n * H + y
  const int y = row % H;
  const float ninf = -__builtin_huge_valf();
  float4 m = make_float4(ninf, ninf, ninf, ninf);
  const float* base = buf + (size_t)(row - y) * W * cs + src_co + c4 * 4;
  for (int dy = -2; dy <= 2; ++dy) {
    const int yy = y + dy;
    if ((unsigned)yy >= (unsigned)H) continue;
    for (int dx = -2; dx <= 2; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      m = max4(m, *reinterpret_cast<const float4*>(base + ((size_t)yy * W + xx) * cs));
    }
  }
  *reinterpret_cast<float4*>(buf + ((size_t)row * W + x) * cs + dst_co + c4 * 4) = m;
}

// ---------------------------------------------------------------------------------
// Nearest resample of an NHWC view into a channel slice of another NHWC buffer:
// F.interpolate(mode='nearest', size=...) + torch.cat of yolo_neck_csp.py:213-219, and
// (Hs==Hd) the plain concat copy of yolo_neck_csp.py:229.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_nearest_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int N, int Hs, int Ws, int Hd, int Wd, int C4,
                                                               int src_cs, int src_co, int dst_cs, int dst_co,
                                                               float hscale, float wscale) {
  const size_t total = (size_t)N * Hd * Wd * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int x = (int)(t % Wd);
    t /= Wd;
    const int y = (int)(t % Hd);
    const int n = (int)(t / Hd);
    // torch nearest: src = min(floor(dst * scale), in - 1) with scale = in/out as float
    int sy = (int)floorf((float)y * hscale);
    int sx = (int)floorf((float)x * wscale);
    sy = sy < Hs - 1 ? sy : Hs - 1;
    sx = sx < Ws - 1 ? sx : Ws - 1;
    const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)(n * Hs + sy) * Ws + sx) * src_cs + src_co + c4 * 4);
    *reinterpret_cast<float4*>(dst + ((size_t)(n * Hd + y) * Wd + x) * dst_cs + dst_co + c4 * 4) = v;
  }
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_mish_fwd(const void* in, void* out, size_t n, int dtype, void* stream) {
  if (n == 0) return YV4_OK;
  YV4_REQUIRE(in && out, "mish_fwd: null pointer");
  YV4_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "mish_fwd: buffers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (dtype) {
    case YV4_F32:
      hipLaunchKernelGGL((mish_fwd_kernel<float, 4>), dim3(ew_grid(n / 4 + 1)), dim3(256), 0, s,
                         (const float*)in, (float*)out, n);
      break;
    case YV4_F16:
      hipLaunchKernelGGL((mish_fwd_kernel<__half, 8>), dim3(ew_grid(n / 8 + 1)), dim3(256), 0, s,
                         (const __half*)in, (__half*)out, n);
      break;
    case YV4_BF16:
      hipLaunchKernelGGL((mish_fwd_kernel<__hip_bfloat16, 8>), dim3(ew_grid(n / 8 + 1)), dim3(256), 0, s,
                         (const __hip_bfloat16*)in, (__hip_bfloat16*)out, n);
      break;
    case YV4_F64:
      hipLaunchKernelGGL(mish_fwd_f64_kernel, dim3(ew_grid(n)), dim3(256), 0, s, (const double*)in, (double*)out, n);
      break;
    default:
      set_error("mish_fwd: unknown dtype %d", dtype);
      return YV4_E_INVALID;
  }
  YV4_CHECK_LAUNCH("mish_fwd");
  return YV4_OK;
}

extern "C" int yv4_mish_bwd(const void* gout, const void* in, void* gin, size_t n, int dtype, void* stream) {
  if (n == 0) return YV4_OK;
  YV4_REQUIRE(gout && in && gin, "mish_bwd: null pointer");
  YV4_REQUIRE((((uintptr_t)in | (uintptr_t)gout | (uintptr_t)gin) & 15) == 0, "mish_bwd: buffers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (dtype) {
    case YV4_F32:
      hipLaunchKernelGGL((mish_bwd_kernel<float, 4>), dim3(ew_grid(n / 4 + 1)), dim3(256), 0, s,
                         (const float*)gout, (const float*)in, (float*)gin, n);
      break;
    case YV4_F16:
      hipLaunchKernelGGL((mish_bwd_kernel<__half, 8>), dim3(ew_grid(n / 8 + 1)), dim3(256), 0, s,
                         (const __half*)gout, (const __half*)in, (__half*)gin, n);
      break;
    case YV4_BF16:
      hipLaunchKernelGGL((mish_bwd_kernel<__hip_bfloat16, 8>), dim3(ew_grid(n / 8 + 1)), dim3(256), 0, s,
                         (const __hip_bfloat16*)gout, (const __hip_bfloat16*)in, (__hip_bfloat16*)gin, n);
      break;
    case YV4_F64:
      hipLaunchKernelGGL(mish_bwd_f64_kernel, dim3(ew_grid(n)), dim3(256), 0, s, (const double*)gout,
                         (const double*)in, (double*)gin, n);
      break;
    default:
      set_error("mish_bwd: unknown dtype %d", dtype);
      return YV4_E_INVALID;
  }
  YV4_CHECK_LAUNCH("mish_bwd");
  return YV4_OK;
}

extern "C" int yv4_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int dst_cstride,
                                int dst_coff, int zero_pad, void* stream) {
  YV4_REQUIRE(src && dst, "nchw_to_nhwc: null pointer");
  YV4_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && zero_pad >= 0, "nchw_to_nhwc: bad shape");
  YV4_REQUIRE(dst_coff >= 0 && dst_coff + C + zero_pad <= dst_cstride, "nchw_to_nhwc: view exceeds pixel stride");
  YV4_REQUIRE(N <= 65535, "nchw_to_nhwc: N > 65535");
  const int HW = H * W;
  if (C + zero_pad == 4 && dst_cstride % 4 == 0 && dst_coff % 4 == 0 && ((uintptr_t)dst & 15) == 0) {
    unsigned gx = (unsigned)((HW + 255) / 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(gx, (unsigned)N), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, dst, C, HW, N, dst_cstride, dst_coff);
    YV4_CHECK_LAUNCH("nchw_to_nhwc");
    return YV4_OK;
  }
  dim3 grid((HW + 63) / 64, (C + zero_pad + 63) / 64, N);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, C, HW,
                     dst_cstride, dst_coff, zero_pad);
  YV4_CHECK_LAUNCH("nchw_to_nhwc");
  return YV4_OK;
}

extern "C" int yv4_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int src_cstride,
                                int src_coff, void* stream) {
  YV4_REQUIRE(src && dst, "nhwc_to_nchw: null pointer");
  YV4_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad shape");
  YV4_REQUIRE(src_coff >= 0 && src_coff + C <= src_cstride, "nhwc_to_nchw: view exceeds pixel stride");
  YV4_REQUIRE(N <= 65535, "nhwc_to_nchw: N > 65535");
  const int HW = H * W;
  dim3 grid((HW + 63) / 64, (C + 63) / 64, N);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, C, HW,
                     src_cstride, src_coff);
  YV4_CHECK_LAUNCH("nhwc_to_nchw");
  return YV4_OK;
}

extern "C" int yv4_spp_pool_fwd(float* buf, int N, int H, int W, int C, int cstride, int coff, void* stream) {
  YV4_REQUIRE(buf, "spp: null pointer");
  YV4_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "spp: bad shape");
  YV4_REQUIRE(C % 4 == 0 && cstride % 4 == 0 && coff % 4 == 0, "spp: C, cstride, coff must be multiples of 4");
  YV4_REQUIRE(coff >= 0 && coff + 4 * C <= cstride, "spp: the 4*C concat slice exceeds the pixel stride");
  YV4_REQUIRE(((uintptr_t)buf & 15) == 0, "spp: buffer must be 16-byte aligned");
  if (H * W <= kSppLdsMaxHW && N <= 65535) {      // the whole map of a channel slice in LDS: one launch (spp_lds.h)
    typedef float f32x4_e __attribute__((ext_vector_type(4)));
    if (int rc = spp_lds_launch<float, f32x4_e, 4>(buf, N, H, W, C, cstride, coff, reinterpret_cast<hipStream_t>(stream), "spp_pool")) return rc;
    YV4_CHECK_LAUNCH("spp_pool");
    return YV4_OK;
  }
  if ((long long)N * H <= 65535) {      // three chained 5x5 pools: x -> mp5 -> mp9 -> mp13
    const dim3 grid((unsigned)((W * (C / 4) + 255) / 256), (unsigned)(N * H));
    for (int k = 0; k < 3; ++k)
      hipLaunchKernelGGL(pool5_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), buf, H, W, C / 4, cstride,
                         coff + k * C, coff + (k + 1) * C);
    YV4_CHECK_LAUNCH("spp_pool");
    return YV4_OK;
  }
  const size_t total = (size_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(spp_pool_kernel, dim3(ew_grid(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), buf, N,
                     H, W, C / 4, cstride, coff, C);
  YV4_CHECK_LAUNCH("spp_pool");
  return YV4_OK;
}

extern "C" int yv4_resample_nearest_fwd(const float* src, float* dst, int N, int Hs, int Ws, int Hd, int Wd, int C,
                                        int src_cstride, int src_coff, int dst_cstride, int dst_coff, void* stream) {
  YV4_REQUIRE(src && dst, "resample: null pointer");
  YV4_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0, "resample: bad shape");
  YV4_REQUIRE(C % 4 == 0 && src_cstride % 4 == 0 && src_coff % 4 == 0 && dst_cstride % 4 == 0 && dst_coff % 4 == 0,
              "resample: channel counts/offsets must be multiples of 4");
  YV4_REQUIRE(src_coff >= 0 && src_coff + C <= src_cstride && dst_coff >= 0 && dst_coff + C <= dst_cstride,
              "resample: view exceeds pixel stride");
  YV4_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "resample: buffers must be 16-byte aligned");
  const size_t total = (size_t)N * Hd * Wd * (C / 4);
  hipLaunchKernelGGL(resample_nearest_kernel, dim3(ew_grid(total)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, dst, N, Hs, Ws, Hd, Wd, C / 4, src_cstride, src_coff,
                     dst_cstride, dst_coff, (float)Hs / (float)Hd, (float)Ws / (float)Wd);
  YV4_CHECK_LAUNCH("resample_nearest");
  return YV4_OK;
}

// ---- the op on HOST memory (ABI 6) ---------------------------------------------------------------------------------
// mmdet/ops/mish_cuda/src/mish.cc:14-33 dispatches on input.is_cuda(): a CPU tensor goes to mish_cpu_kernel
// (src/kernel/mish_cpu.cc:6-29), a plain loop over mish.h:16-29 under AT_DISPATCH_ALL_TYPES -- float and double; Half and
// BFloat16 are not dispatched on the CPU and raise there, so they are refused here.  One property of that loop is part
// of its arithmetic: mish.h is included INSIDE namespace mish_cpu_kernel, where the unqualified exp / log1p / tanh of a
// float argument bind to the C library's double functions -- the float forward is evaluated in double and rounded once,
// the float backward rounds at each of its named `const scalar_t` values.  The loops below do exactly that, so the
// op's CPU results are the reference's to the bit (tests/test_abi.py, against the fixture its unmodified source made).
// Single-threaded, like the reference's loop.  Not a path of any plan or train step.
namespace {
inline float mish_fwd_host_f(float x) {
  return (float)((double)x * ::tanh(x < 20.f ? ::log1p(::exp((double)x)) : (double)x));
}
inline float mish_bwd_host_f(float go, float x) {
  const float sp = (float)(x < 20.f ? ::log1p(::exp((double)x)) : (double)x);
  const float grad_sp = (float)(1 - ::exp(-(double)sp));
  const float tsp = (float)::tanh((double)sp);
  const float grad_tsp = (1 - tsp * tsp) * grad_sp;
  const float grad = x * grad_tsp + tsp;
  return go * grad;
}
inline double mish_fwd_host_d(double x) { return x * ::tanh(x < 20.0 ? ::log1p(::exp(x)) : x); }
inline double mish_bwd_host_d(double go, double x) {
  const double sp = x < 20.0 ? ::log1p(::exp(x)) : x;
  const double grad_sp = 1.0 - ::exp(-sp);
  const double tsp = ::tanh(sp);
  const double grad_tsp = (1.0 - tsp * tsp) * grad_sp;
  const double grad = x * grad_tsp + tsp;
  return go * grad;
}
}  // namespace

// (no contraction: `x * grad_tsp + tsp` is a multiply and an add in the reference's build)
#pragma clang fp contract(off)
extern "C" int yv4_mish_fwd_host(const void* in, void* out, size_t n, int dtype) {
  if (n == 0) return YV4_OK;
  YV4_REQUIRE(in && out, "mish_fwd_host: null pointer");
  if (dtype == YV4_F32) {
    const float* x = (const float*)in; float* y = (float*)out;
    for (size_t i = 0; i < n; ++i) y[i] = mish_fwd_host_f(x[i]);
  } else if (dtype == YV4_F64) {
    const double* x = (const double*)in; double* y = (double*)out;
    for (size_t i = 0; i < n; ++i) y[i] = mish_fwd_host_d(x[i]);
  } else {
    set_error("mish_fwd_host: \"mish_cpu_kernel\" not implemented for this dtype (%d): the reference's CPU loop takes float "
              "and double", dtype);
    return YV4_E_UNSUPPORTED;
  }
  return YV4_OK;
}

extern "C" int yv4_mish_bwd_host(const void* gout, const void* in, void* gin, size_t n, int dtype) {
  if (n == 0) return YV4_OK;
  YV4_REQUIRE(gout && in && gin, "mish_bwd_host: null pointer");
  if (dtype == YV4_F32) {
    const float* g = (const float*)gout; const float* x = (const float*)in; float* y = (float*)gin;
    for (size_t i = 0; i < n; ++i) y[i] = mish_bwd_host_f(g[i], x[i]);
  } else if (dtype == YV4_F64) {
    const double* g = (const double*)gout; const double* x = (const double*)in; double* y = (double*)gin;
    for (size_t i = 0; i < n; ++i) y[i] = mish_bwd_host_d(g[i], x[i]);
  } else {
    set_error("mish_bwd_host: \"mish_backward_cpu_kernel\" not implemented for this dtype (%d): the reference's CPU loop takes "
              "float and double", dtype);
    return YV4_E_UNSUPPORTED;
  }
  return YV4_OK;
}
