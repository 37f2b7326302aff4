// Shared pieces of the fp32 fused convolution kernels (conv_mfma_f32.hip: the implicit-GEMM tiles on 32x32x2 MFMAs;
// conv3x3_wide_f32.hip: the wide-tile 3x3 kernel on 16x16x4 MFMAs): argument block, scattered-row map, LDS-DMA helpers.
#pragma once
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
  // scattered output (sub-pixel / parity convolutions of the stride-2 data gradient): output pixel
  // (n, ho, wo) is stored at row ((n*ys_H + ho*ys_sh + ys_oh)*ys_W + wo*ys_sw + ys_ow) of y
  int ys_on, ys_H, ys_W, ys_sh, ys_sw, ys_oh, ys_ow;
  double* stats;   // training: [YV4_STATS_REPLICAS][sum (Cout) | sum of squares (Cout)] of the outputs, or null
  FastDiv fd_hw, fd_wo;   // m / (Ho*Wo), r / Wo (LDS-DMA kernels; set by launch_conv_dma)
  // split-K (LDS-DMA kernels, single-image plans): workgroup (tile, split) reduces K slices
  // [split * ks_slices, ...) and stores its RAW partial tile into slab `split` of ws ([ksplit][M][ws_cs]);
  // splitk_finish_kernel adds the slabs in slab order and applies the epilogue.  ksplit <= 1: off.
  int ksplit = 0, ks_slices = 0, ws_cs = 0;
  float* ws = nullptr;
  FastDiv fd_taps, fd_kw;   // slice -> (chunk, tap), tap -> (kh, kw) at a split's first slice
};

__device__ __forceinline__ int64_t out_row(const ConvArgs& p, int m) {
  if (!p.ys_on) return m;
  const int hw = p.Ho * p.Wo;
  const int n = m / hw;
  const int r = m - n * hw;
  const int ho = r / p.Wo;
  const int wo = r - ho * p.Wo;
  return ((int64_t)n * p.ys_H + ho * p.ys_sh + p.ys_oh) * p.ys_W + wo * p.ys_sw + p.ys_ow;
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// One LDS-DMA wave-instruction: lane l's 16 bytes at (descriptor base + voff + soff) land at
// LDS byte address lds_addr + 16*l.  Issued through inline asm on purpose: hipcc would
// otherwise wait vmcnt(0) before the next ds_read of ANY LDS address (it cannot tell the two
// halves of the double buffer apart), exposing the whole memory latency every K step.  The
// kernel counts these loads itself: s_waitcnt vmcnt(0) + s_barrier before the slice is read.
__device__ __forceinline__ void lds_dma16(u32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

__device__ __forceinline__ u32x4_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  u32x4_t v;
  v.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  v.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  v.z = __builtin_amdgcn_readfirstlane(bytes);
  v.w = 0x00020000u;
  return v;
}

}  // namespace yv4
