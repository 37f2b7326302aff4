// Test-time input pipeline of the YOLOv4 configs in one kernel per image:
//   Resize(keep_ratio=True) -> Pad(size_divisor) -> Normalize(mean, std, to_rgb) -> ImageToTensor
// (configs/yolov4/yolov4l_coco_mosaic.py:70-84; mmdet/datasets/pipelines/transforms.py Resize / Pad / Normalize;
// the arithmetic itself lives in mmcv / OpenCV, which are third party and absent from the build image:
// PARITY UNPINNED -- oracle/preprocess_oracle.py restates OpenCV's documented 8-bit INTER_LINEAR (fixed point,
// modules/imgproc/src/resize.cpp) and mmcv.imnormalize's float32 subtract / multiply, and this kernel is tested
// bit for bit against that restatement only).
//
// One thread per output pixel of the padded (Hp x Wp) image: inside the resized extent it interpolates the 8-bit
// source with OpenCV's integer arithmetic (coefficients scaled by 2^11, two-stage rounding), outside it takes the
// pad value; then (v - mean) * (1 / std) per channel in fp32, channel order swapped if to_rgb; written planar
// (3, Hp, Wp) fp32 into the image's slot of the batch tensor (NCHW, what SingleStageDetector.simple_test takes).
#include "yv4_common.h"

namespace yv4 {

struct PreArgs {
  const uint8_t* src; int sh, sw, s_pitch;      // source image, HWC, 3 channels, row pitch in bytes
  float* dst; int Hp, Wp; long long plane;      // destination planes (channel stride `plane` elements)
  int nh, nw;                                   // resized extent
  double inv_sx, inv_sy;                        // source / destination size ratios (OpenCV: scale_x = 1. / inv_scale_x)
  float mean[3], stdinv[3];
  int to_rgb, pad_val, pad_first;               // pad_first: Pad precedes Normalize (the pad value is normalised too)
};

// OpenCV resize, INTER_LINEAR, 8-bit: sample position and 11-bit coefficients of one axis
__device__ __forceinline__ void lin_coef(int d, double inv_scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
  float f = (float)((d + 0.5) * inv_scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }          // (OpenCV clamps the second tap to the last pixel)
  s0 = s;
  s1 = s + 1 < ssize ? s + 1 : s;
  a0 = (int)rintf((1.f - f) * 2048.f);
  a1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void letterbox_u8_kernel(PreArgs p) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= p.Wp || y >= p.Hp) return;
  int v[3] = {p.pad_val, p.pad_val, p.pad_val};
  const bool inside = x < p.nw && y < p.nh;
  if (inside) {
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    lin_coef(x, p.inv_sx, p.sw, x0, x1, ax0, ax1);
    lin_coef(y, p.inv_sy, p.sh, y0, y1, ay0, ay1);
    const uint8_t* r0 = p.src + (size_t)y0 * p.s_pitch;
    const uint8_t* r1 = p.src + (size_t)y1 * p.s_pitch;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;       // horizontal pass, 11 fractional bits
      const int h1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
      // vertical pass: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16), rounded: (+ 2) >> 2
      const int r = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
      v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = p.to_rgb ? 2 - c : c;                     // output channel c reads source channel sc
    float o;
    if (inside || p.pad_first) o = ((float)v[sc] - p.mean[c]) * p.stdinv[c];
    else o = (float)p.pad_val;
    p.dst[(size_t)c * p.plane + (size_t)y * p.Wp + x] = o;
  }
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_letterbox_u8(const uint8_t* src, int src_h, int src_w, int src_pitch, float* dst, int Hp, int Wp,
                                int64_t plane_stride, int new_h, int new_w, const float* mean3, const float* std3,
                                int to_rgb, int pad_val, int pad_before_normalize, void* stream) {
  YV4_REQUIRE(src && dst && mean3 && std3, "letterbox: null pointer");
  YV4_REQUIRE(src_h > 0 && src_w > 0 && src_pitch >= 3 * src_w && Hp > 0 && Wp > 0 && plane_stride >= (int64_t)Hp * Wp,
              "letterbox: bad geometry");
  YV4_REQUIRE(new_h > 0 && new_w > 0 && new_h <= Hp && new_w <= Wp, "letterbox: the resized image exceeds the padded one");
  YV4_REQUIRE(pad_val >= 0 && pad_val <= 255, "letterbox: pad value must be an 8-bit value");
  PreArgs a;
  a.src = src; a.sh = src_h; a.sw = src_w; a.s_pitch = src_pitch;
  a.dst = dst; a.Hp = Hp; a.Wp = Wp; a.plane = plane_stride;
  a.nh = new_h; a.nw = new_w;
  a.inv_sx = 1.0 / ((double)new_w / (double)src_w);     // OpenCV: scale_x = 1. / ((double)dsize.width / ssize.width)
  a.inv_sy = 1.0 / ((double)new_h / (double)src_h);
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean3[c];
    a.stdinv[c] = (float)(1.0 / (double)std3[c]);
  }
  a.to_rgb = to_rgb ? 1 : 0; a.pad_val = pad_val; a.pad_first = pad_before_normalize ? 1 : 0;
  hipLaunchKernelGGL(letterbox_u8_kernel, dim3((unsigned)((Wp + 63) / 64), (unsigned)((Hp + 3) / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  YV4_CHECK_LAUNCH("letterbox_u8");
  return YV4_OK;
}
