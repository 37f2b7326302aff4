// SPP max-pools (darknetcsp.py:176-181,203-206,222-226: cat([x, mp5(x), mp9(x), mp13(x)])) of a small map as ONE launch.
// A 19 x 19 or 13 x 13 map of a 64-byte channel slice (16 fp32 / 32 sixteen-bit channels) fits the LDS twice over: a
// workgroup owns (image, slice), reads x once, and runs the three CHAINED 5 x 5 pools (max is associative and idempotent:
// mp9 = mp5 o mp5, mp13 = mp5 o mp9 with windows clipped at the border -- exact) as separable row / column passes between
// two LDS planes, storing each pool's result into its slice of the concat buffer.  The three chained launches this
// replaces read 25 taps per output from L2 / HBM: ~10 x the map per launch (40 us each at 19 x 19 x 512 x 32 for a 5 us
// copy).  Maps above kSppLdsMaxHW pixels keep the chained launches.
#ifndef YV4_SPP_LDS_H_
#define YV4_SPP_LDS_H_
#include "yv4_common.h"

namespace yv4 {

constexpr int kSppLdsMaxHW = 512;          // 2 planes x 512 pixels x 64 bytes = 64 KB

// T: element, V: 16-byte vector of VE elements; a pixel's slice = 4 vectors
template <typename T, typename V, int VE>
__global__ __launch_bounds__(256) void spp_lds_kernel(T* __restrict__ buf, int H, int W, int cs, int co, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem_spp[];
  const int HW = H * W;
  V* A = reinterpret_cast<V*>(smem_spp);
  V* B = A + HW * 4;
  const int n = blockIdx.y;
  const int c0 = blockIdx.x * (4 * VE);
  const int nv = min(4, (C - c0) / VE);              // whole vectors of this slice (C % VE == 0)
  const int items = HW * 4;
  T* img = buf + (size_t)n * HW * cs + co + c0;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int p = i >> 2, v = i & 3;
    if (v < nv) A[i] = *reinterpret_cast<const V*>(img + (size_t)p * cs + v * VE);
  }
  __syncthreads();
  for (int k = 1; k <= 3; ++k) {
    for (int i = threadIdx.x; i < items; i += 256) {          // rows: A -> B
      const int p = i >> 2, v = i & 3;
      if (v >= nv) continue;
      const int y = p / W, x = p - y * W;
      float m[VE];
      const V c = A[i];
#pragma unroll
      for (int u = 0; u < VE; ++u) m[u] = (float)c[u];
#pragma unroll
      for (int d = -2; d <= 2; ++d) {
        if (d == 0 || (unsigned)(x + d) >= (unsigned)W) continue;
        const V t = A[i + 4 * d];
#pragma unroll
        for (int u = 0; u < VE; ++u) m[u] = fmaxf(m[u], (float)t[u]);
      }
      V o;
#pragma unroll
      for (int u = 0; u < VE; ++u) o[u] = (T)m[u];
      B[i] = o;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < items; i += 256) {          // columns: B -> A, and out
      const int p = i >> 2, v = i & 3;
      if (v >= nv) continue;
      const int y = p / W;
      float m[VE];
      const V c = B[i];
#pragma unroll
      for (int u = 0; u < VE; ++u) m[u] = (float)c[u];
#pragma unroll
      for (int d = -2; d <= 2; ++d) {
        if (d == 0 || (unsigned)(y + d) >= (unsigned)H) continue;
        const V t = B[i + 4 * d * W];
#pragma unroll
        for (int u = 0; u < VE; ++u) m[u] = fmaxf(m[u], (float)t[u]);
      }
      V o;
#pragma unroll
      for (int u = 0; u < VE; ++u) o[u] = (T)m[u];
      A[i] = o;
      *reinterpret_cast<V*>(img + (size_t)k * C + (size_t)p * cs + v * VE) = o;
    }
    __syncthreads();
  }
}

template <typename T, typename V, int VE>
static int spp_lds_launch(T* buf, int N, int H, int W, int C, int cs, int co, hipStream_t s, const char* who) {
  const size_t lds = (size_t)2 * H * W * 64;
  auto kern = spp_lds_kernel<T, V, VE>;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), (size_t)2 * kSppLdsMaxHW * 64, who)) return rc;
  const dim3 grid((unsigned)((C + 4 * VE - 1) / (4 * VE)), (unsigned)N);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, buf, H, W, cs, co, C);
  return YV4_OK;
}

}  // namespace yv4
#endif
