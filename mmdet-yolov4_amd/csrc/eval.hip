// Detection-evaluation kernels: box IoU in COCO convention and the greedy COCO matching of detections
// to ground truth -- the reference's two Cython ops
//   mmdet/ops/eval_utils/iou/iou_coco.pyx:8-56      iou_coco(det_boxes, gt_boxes, is_crowd)
//   mmdet/ops/eval_utils/match/match_coco.pyx:8-57  match_coco(iou_mat, iou_thrs, is_ignore, is_crowd)
// (called per image and class from core/evaluation/mean_ap_flexible.py:19-37), batched: one launch
// evaluates every (image, class) problem of a dataset.  Problem p owns detections
// [det_off[p], det_off[p+1]), ground truths [gt_off[p], gt_off[p+1]) and the row-major
// (num_det x num_gt) IoU block at iou_off[p].
// Arithmetic is the reference's fp32 expression order (compiled with -ffp-contract=off): bit-exact.
#include "yv4_common.h"

namespace yv4 {

__device__ __forceinline__ int find_problem(const int64_t* __restrict__ off, int P, int64_t i) {
  int lo = 0, hi = P;       // invariant: off[lo] <= i < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void iou_coco_kernel(const float* __restrict__ det, const float* __restrict__ gt,
                                                       const uint8_t* __restrict__ is_crowd,
                                                       const int64_t* __restrict__ det_off,
                                                       const int64_t* __restrict__ gt_off,
                                                       const int64_t* __restrict__ iou_off, int P,
                                                       float* __restrict__ iou) {
  const int64_t total = iou_off[P];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int p = find_problem(iou_off, P, i);
    const int64_t ng = gt_off[p + 1] - gt_off[p];
    const int64_t local = i - iou_off[p];
    const int64_t d = det_off[p] + local / ng;
    const int64_t g = gt_off[p] + local % ng;
    const float4 db = reinterpret_cast<const float4*>(det)[d];
    const float4 gb = reinterpret_cast<const float4*>(gt)[g];
    const float tlx = fmaxf(db.x, gb.x), tly = fmaxf(db.y, gb.y);
    const float brx = fminf(db.z, gb.z), bry = fminf(db.w, gb.w);
    float v = 0.f;
    if (!(tlx >= brx || tly >= bry)) {
      const float inter = (brx - tlx) * (bry - tly);
      const float darea = (db.z - db.x) * (db.w - db.y);
      float uni;
      if (is_crowd[g]) {
        uni = darea;
      } else {
        const float garea = (gb.z - gb.x) * (gb.w - gb.y);
        uni = darea + garea - inter;
      }
      if (uni <= 0.f) uni = 1e-7f;
      v = inter / uni;
    }
    iou[i] = v;
  }
}

// One thread per (problem, IoU threshold): the loop nest of match_coco.pyx:27-55 verbatim.
__global__ __launch_bounds__(64) void match_coco_kernel(const float* __restrict__ iou, const int64_t* __restrict__ det_off,
                                                        const int64_t* __restrict__ gt_off,
                                                        const int64_t* __restrict__ iou_off,
                                                        const float* __restrict__ thrs, int nt,
                                                        const uint8_t* __restrict__ is_ignore,
                                                        const uint8_t* __restrict__ is_crowd, int P,
                                                        uint8_t* __restrict__ gt_matched, int32_t* __restrict__ matched) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)P * nt) return;
  const int p = (int)(id / nt);
  const int t = (int)(id % nt);
  const int64_t nd = det_off[p + 1] - det_off[p];
  const int64_t ng = gt_off[p + 1] - gt_off[p];
  const float* m = iou + iou_off[p];
  const uint8_t* ign = is_ignore + gt_off[p];
  const uint8_t* crowd = is_crowd + gt_off[p];
  uint8_t* used = gt_matched + (gt_off[p] * nt + (int64_t)t * ng);       // [nt][ng] block of the problem
  int32_t* out = matched + (det_off[p] * nt + (int64_t)t * nd);          // [nt][nd] block of the problem
  for (int64_t g = 0; g < ng; ++g) used[g] = 0;
  const float thr = thrs[t];
  for (int64_t d = 0; d < nd; ++d) {
    float best = thr, best_ignore = thr;
    int32_t mg = -1;
    for (int64_t g = 0; g < ng; ++g) {
      if (used[g] && !crowd[g]) continue;
      if (mg > -1 && !ign[mg] && ign[g]) continue;      // matched to a regular gt and now on an ignore gt
      const float need = ign[g] ? best_ignore : best;
      const float v = m[d * ng + g];
      if (v < need) continue;
      if (ign[g]) best_ignore = v; else best = v;
      mg = (int32_t)g;
    }
    if (mg != -1) used[mg] = 1;
    out[d] = mg;
  }
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_iou_coco_batched(const float* det, const float* gt, const uint8_t* is_crowd, const int64_t* det_off,
                                    const int64_t* gt_off, const int64_t* iou_off, int P, int64_t total_pairs,
                                    float* iou, void* stream) {
  YV4_REQUIRE(P > 0 && det_off && gt_off && iou_off, "iou_coco: bad problem table");
  YV4_REQUIRE(total_pairs >= 0, "iou_coco: negative pair count");
  if (total_pairs == 0) return YV4_OK;
  YV4_REQUIRE(det && gt && is_crowd && iou, "iou_coco: null pointer");
  YV4_REQUIRE((((uintptr_t)det | (uintptr_t)gt) & 15) == 0, "iou_coco: boxes must be 16-byte aligned");
  long long blocks = (total_pairs + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(iou_coco_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), det, gt,
                     is_crowd, det_off, gt_off, iou_off, P, iou);
  YV4_CHECK_LAUNCH("iou_coco");
  return YV4_OK;
}

extern "C" int yv4_match_coco_batched(const float* iou, const int64_t* det_off, const int64_t* gt_off,
                                      const int64_t* iou_off, const float* iou_thrs, int num_thrs,
                                      const uint8_t* is_ignore, const uint8_t* is_crowd, int P, uint8_t* work,
                                      int32_t* matched, void* stream) {
  YV4_REQUIRE(P > 0 && det_off && gt_off && iou_off && iou_thrs && num_thrs > 0, "match_coco: bad problem table");
  YV4_REQUIRE(is_ignore && is_crowd && work && matched, "match_coco: null pointer");
  const long long n = (long long)P * num_thrs;
  hipLaunchKernelGGL(match_coco_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0,
                     reinterpret_cast<hipStream_t>(stream), iou, det_off, gt_off, iou_off, iou_thrs, num_thrs, is_ignore,
                     is_crowd, P, work, matched);
  YV4_CHECK_LAUNCH("match_coco");
  return YV4_OK;
}
