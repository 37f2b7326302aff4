// 16-bit (fp16 / bf16) forms of the non-conv kernels of the fused path: the layout adaptors at
// the module boundary (the reference's tensors are fp32 NCHW; the 16-bit path keeps NHWC fp16 / bf16
// inside) and the SPP max-pools (darknetcsp.py:176-181,203-206,222-226).  The nearest resample /
// concat copy needs no 16-bit kernel: a 16-bit NHWC view with C % 8 == 0 is byte-identical to an
// fp32 view with C/2 channels, so the host calls yv4_resample_nearest_fwd with halved channel
// arguments.
#include "yv4_common.h"
#include "spp_lds.h"

namespace yv4 {

typedef __bf16 bf16x8_e __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_e __attribute__((ext_vector_type(8)));

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_h16_kernel(const float* __restrict__ src, T* __restrict__ dst, int C,
                                                               int HW, int dst_cs, int dst_co, int cpad) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63;
  const int ty = threadIdx.x >> 6;
  const float* s = src + (size_t)n * C * HW;
  for (int c = ty; c < 64; c += 4) {
    const int ch = c0 + c;
    const int px = p0 + tx;
    tile[c][tx] = (ch < C && px < HW) ? s[(size_t)ch * HW + px] : 0.f;
  }
  __syncthreads();
  T* d = dst + (size_t)n * HW * dst_cs + dst_co;
  for (int pp = ty; pp < 64; pp += 4) {
    const int px = p0 + pp;
    const int ch = c0 + tx;
    if (px < HW && ch < C + cpad) d[(size_t)px * dst_cs + ch] = (T)tile[tx][pp];
  }
}

// Small-C form (the 3-channel image padded to NV 16-byte chunks, C + pad == 8 NV): one thread per pixel reads its C
// planar fp32 values (coalesced per plane) and writes 8 NV sixteen-bit values in NV stores; grid (pixel blocks, image).
template <typename T, typename V8, int NV>
__global__ __launch_bounds__(256) void nchw_to_nhwc8_h16_kernel(const float* __restrict__ src, T* __restrict__ dst, int C,
                                                                int HW, int dst_cs, int dst_co) {
  const int n = blockIdx.y;
  const float* s0 = src + (size_t)n * C * HW;
  T* d0 = dst + (size_t)n * HW * dst_cs + dst_co;
  for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += gridDim.x * 256) {
    V8 o[NV];
#pragma unroll
    for (int c = 0; c < 8 * NV; ++c) o[c >> 3][c & 7] = (T)(c < C ? s0[(size_t)c * HW + px] : 0.f);
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<V8*>(d0 + (size_t)px * dst_cs + 8 * v) = o[v];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_h16_kernel(const T* __restrict__ src, float* __restrict__ dst, int C,
                                                               int HW, int src_cs, int src_co) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63;
  const int ty = threadIdx.x >> 6;
  const T* s = src + (size_t)n * HW * src_cs + src_co;
  for (int pp = ty; pp < 64; pp += 4) {
    const int px = p0 + pp;
    const int ch = c0 + tx;
    tile[pp][tx] = (px < HW && ch < C) ? (float)s[(size_t)px * src_cs + ch] : 0.f;
  }
  __syncthreads();
  float* d = dst + (size_t)n * C * HW;
  for (int c = ty; c < 64; c += 4) {
    const int ch = c0 + c;
    const int px = p0 + tx;
    if (ch < C && px < HW) d[(size_t)ch * HW + px] = tile[tx][c];
  }
}

// MaxPool2d(5, 1, 2) of one channel slice into another; chained three times it gives the 5 / 9 / 13 pools exactly
// (see pool5_kernel in elementwise.hip).
template <typename T, typename V8>
__global__ __launch_bounds__(256) void pool5_h16_kernel(T* __restrict__ buf, int H, int W, int C8, int cs, int src_co,
                                                        int dst_co) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * C8) return;
  const int x = idx / C8, c8 = idx - x * C8;
  const int row = blockIdx.y;
  const int y = row % H;
  const float ninf = -__builtin_huge_valf();
  float m[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) m[u] = ninf;
  const T* base = buf + (size_t)(row - y) * W * cs + src_co + c8 * 8;
  for (int dy = -2; dy <= 2; ++dy) {
    const int yy = y + dy;
    if ((unsigned)yy >= (unsigned)H) continue;
    for (int dx = -2; dx <= 2; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      const V8 v = *reinterpret_cast<const V8*>(base + ((size_t)yy * W + xx) * cs);
#pragma unroll
      for (int u = 0; u < 8; ++u) m[u] = fmaxf(m[u], (float)v[u]);
    }
  }
  V8 o;
#pragma unroll
  for (int u = 0; u < 8; ++u) o[u] = (T)m[u];
  *reinterpret_cast<V8*>(buf + ((size_t)row * W + x) * cs + dst_co + c8 * 8) = o;
}

// One thread owns (n, y, x, 8 channels): a 16-byte load per tap, maxima kept in fp32 (exact for
// values that are already fp16 / bf16), three 16-byte stores.
template <typename T, typename V8>
__global__ __launch_bounds__(256) void spp_pool_h16_kernel(T* __restrict__ buf, int N, int H, int W, int C8, int cs, int co,
                                                           int C) {
  const size_t total = (size_t)N * H * W * C8;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float ninf = -__builtin_huge_valf();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    size_t t = i / C8;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const T* base = buf + (size_t)n * H * W * cs + co + c8 * 8;
    float m5[8], m9[8], m13[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) m5[u] = m9[u] = m13[u] = ninf;
    for (int dy = -6; dy <= 6; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const int ady = dy < 0 ? -dy : dy;
      for (int dx = -6; dx <= 6; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const int adx = dx < 0 ? -dx : dx;
        const int rad = ady > adx ? ady : adx;
        const V8 v = *reinterpret_cast<const V8*>(base + ((size_t)yy * W + xx) * cs);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float f = (float)v[u];
          m13[u] = fmaxf(m13[u], f);
          if (rad <= 4) m9[u] = fmaxf(m9[u], f);
          if (rad <= 2) m5[u] = fmaxf(m5[u], f);
        }
      }
    }
    T* o = buf + ((size_t)(n * H + y) * W + x) * cs + co + c8 * 8;
    V8 o5, o9, o13;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      o5[u] = (T)m5[u];
      o9[u] = (T)m9[u];
      o13[u] = (T)m13[u];
    }
    *reinterpret_cast<V8*>(o + C) = o5;
    *reinterpret_cast<V8*>(o + 2 * C) = o9;
    *reinterpret_cast<V8*>(o + 3 * C) = o13;
  }
}

static inline unsigned ew_grid_h(size_t work_items) {
  size_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (unsigned)g;
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_nchw_to_nhwc_h16(const float* src, void* dst, int N, int C, int H, int W, int dst_cstride, int dst_coff,
                                    int zero_pad, int dtype, void* stream) {
  YV4_REQUIRE(src && dst, "nchw_to_nhwc_h16: null pointer");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "nchw_to_nhwc_h16: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && zero_pad >= 0, "nchw_to_nhwc_h16: bad shape");
  YV4_REQUIRE(dst_coff >= 0 && dst_coff + C + zero_pad <= dst_cstride, "nchw_to_nhwc_h16: view exceeds pixel stride");
  YV4_REQUIRE(N <= 65535, "nchw_to_nhwc_h16: N > 65535");
  const int HW = H * W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((C + zero_pad == 8 || C + zero_pad == 16) && dst_cstride % 8 == 0 && dst_coff % 8 == 0 && ((uintptr_t)dst & 15) == 0) {
    unsigned gx = (unsigned)((HW + 255) / 256);
    if (gx > 4096) gx = 4096;
    const bool two = C + zero_pad == 16;
    const dim3 grid(gx, (unsigned)N);
    if (dtype == YV4_BF16) {
      if (two) hipLaunchKernelGGL((nchw_to_nhwc8_h16_kernel<__bf16, bf16x8_e, 2>), grid, dim3(256), 0, s, src, (__bf16*)dst, C, HW, dst_cstride, dst_coff);
      else hipLaunchKernelGGL((nchw_to_nhwc8_h16_kernel<__bf16, bf16x8_e, 1>), grid, dim3(256), 0, s, src, (__bf16*)dst, C, HW, dst_cstride, dst_coff);
    } else {
      if (two) hipLaunchKernelGGL((nchw_to_nhwc8_h16_kernel<_Float16, f16x8_e, 2>), grid, dim3(256), 0, s, src, (_Float16*)dst, C, HW, dst_cstride, dst_coff);
      else hipLaunchKernelGGL((nchw_to_nhwc8_h16_kernel<_Float16, f16x8_e, 1>), grid, dim3(256), 0, s, src, (_Float16*)dst, C, HW, dst_cstride, dst_coff);
    }
    YV4_CHECK_LAUNCH("nchw_to_nhwc_h16");
    return YV4_OK;
  }
  dim3 grid((HW + 63) / 64, (C + zero_pad + 63) / 64, N);
  if (dtype == YV4_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_h16_kernel<__bf16>, grid, dim3(256), 0, s, src, (__bf16*)dst, C, HW, dst_cstride,
                       dst_coff, zero_pad);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_h16_kernel<_Float16>, grid, dim3(256), 0, s, src, (_Float16*)dst, C, HW, dst_cstride,
                       dst_coff, zero_pad);
  YV4_CHECK_LAUNCH("nchw_to_nhwc_h16");
  return YV4_OK;
}

extern "C" int yv4_nhwc_to_nchw_h16(const void* src, float* dst, int N, int C, int H, int W, int src_cstride, int src_coff,
                                    int dtype, void* stream) {
  YV4_REQUIRE(src && dst, "nhwc_to_nchw_h16: null pointer");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "nhwc_to_nchw_h16: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw_h16: bad shape");
  YV4_REQUIRE(src_coff >= 0 && src_coff + C <= src_cstride, "nhwc_to_nchw_h16: view exceeds pixel stride");
  YV4_REQUIRE(N <= 65535, "nhwc_to_nchw_h16: N > 65535");
  const int HW = H * W;
  dim3 grid((HW + 63) / 64, (C + 63) / 64, N);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == YV4_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_h16_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)src, dst, C, HW, src_cstride,
                       src_coff);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_h16_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)src, dst, C, HW,
                       src_cstride, src_coff);
  YV4_CHECK_LAUNCH("nhwc_to_nchw_h16");
  return YV4_OK;
}

extern "C" int yv4_spp_pool_fwd_h16(void* buf, int N, int H, int W, int C, int cstride, int coff, int dtype, void* stream) {
  YV4_REQUIRE(buf, "spp_h16: null pointer");
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "spp_h16: dtype must be YV4_F16 or YV4_BF16");
  YV4_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "spp_h16: bad shape");
  YV4_REQUIRE(C % 8 == 0 && cstride % 8 == 0 && coff % 8 == 0, "spp_h16: C, cstride and coff must be multiples of 8");
  YV4_REQUIRE(coff >= 0 && coff + 4 * C <= cstride, "spp_h16: the 4C-channel concat view exceeds the pixel stride");
  YV4_REQUIRE(((uintptr_t)buf & 15) == 0, "spp_h16: buffer must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (H * W <= kSppLdsMaxHW && N <= 65535) {      // the whole map of a channel slice in LDS: one launch (spp_lds.h)
    if (int rc = dtype == YV4_BF16 ? spp_lds_launch<__bf16, bf16x8_e, 8>((__bf16*)buf, N, H, W, C, cstride, coff, s, "spp_pool_h16")
                                   : spp_lds_launch<_Float16, f16x8_e, 8>((_Float16*)buf, N, H, W, C, cstride, coff, s, "spp_pool_h16"))
      return rc;
    YV4_CHECK_LAUNCH("spp_pool_h16");
    return YV4_OK;
  }
  if ((long long)N * H <= 65535) {      // three chained 5x5 pools
    const dim3 grid((unsigned)((W * (C / 8) + 255) / 256), (unsigned)(N * H));
    for (int k = 0; k < 3; ++k) {
      if (dtype == YV4_BF16)
        hipLaunchKernelGGL((pool5_h16_kernel<__bf16, bf16x8_e>), grid, dim3(256), 0, s, (__bf16*)buf, H, W, C / 8, cstride,
                           coff + k * C, coff + (k + 1) * C);
      else
        hipLaunchKernelGGL((pool5_h16_kernel<_Float16, f16x8_e>), grid, dim3(256), 0, s, (_Float16*)buf, H, W, C / 8, cstride,
                           coff + k * C, coff + (k + 1) * C);
    }
    YV4_CHECK_LAUNCH("spp_pool_h16");
    return YV4_OK;
  }
  const size_t total = (size_t)N * H * W * (C / 8);
  if (dtype == YV4_BF16)
    hipLaunchKernelGGL((spp_pool_h16_kernel<__bf16, bf16x8_e>), dim3(ew_grid_h(total)), dim3(256), 0, s, (__bf16*)buf, N, H, W,
                       C / 8, cstride, coff, C);
  else
    hipLaunchKernelGGL((spp_pool_h16_kernel<_Float16, f16x8_e>), dim3(ew_grid_h(total)), dim3(256), 0, s, (_Float16*)buf, N, H,
                       W, C / 8, cstride, coff, C);
  YV4_CHECK_LAUNCH("spp_pool_h16");
  return YV4_OK;
}
