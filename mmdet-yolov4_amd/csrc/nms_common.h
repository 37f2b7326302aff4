// Device helpers shared by the NMS kernels (postproc.hip, nms_split.hip).
// Translation units including this header must be built with -ffp-contract=off.
#pragma once
#include "yv4_common.h"

namespace yv4 {

// ---- order-preserving float <-> uint32 (descending score = ascending key) ---------
__device__ __forceinline__ uint32_t score_to_key(float s) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order of floats
  return ~u;                                       // descending
}
__device__ __forceinline__ float key_to_score(uint32_t k) {
  uint32_t u = ~k;
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

// The suppression predicate, in the two forms mmcv-full 1.3.x ships (the reference's call site is
// core/post_processing/bbox_nms.py:84; mmcv itself is third party and absent):
//   form 0 (YV4_NMS_IOU_DIV, default)  inter / (Sa + Sb - inter) > thr     mmcv's CPU kernel (nms_cpu)
//   form 1 (YV4_NMS_IOU_MUL)           inter > thr * (Sa + Sb - inter)     mmcv's CUDA kernel (devIoU)
// They differ only where the fp32 rounding of the quotient / the product crosses thr (tests/golden/nms_boundary.npz
// holds such pairs).  `form` is uniform over the launch.
__device__ __forceinline__ bool iou_gt(const float4 bi, const float ai, const float4 bj, const float aj, const float thr,
                                       const int form) {
  const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
  const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  // disjoint boxes (most pairs), division form: 0 / uni is +-0 or NaN, none of which is > thr >= 0 -- the same answer
  // without the division.  (The product form keeps its expression: thr * uni is negative for a malformed box.)
  if (!form && !(inter > 0.f) && thr >= 0.f) return false;
  const float uni = ai + aj - inter;
  if (form) return inter > thr * uni;
  const float ovr = inter / uni;
  return ovr > thr;
}

int nms_iou_form();   // api.hip: the process-wide form set by yv4_nms_set_iou_form


}  // namespace yv4
