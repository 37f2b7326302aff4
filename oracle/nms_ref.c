/*
 * oracle/nms_ref.c -- CPU ORACLE (test infrastructure, not the product).
 *
 * Greedy non-maximum suppression as mmcv-full 1.3.x's `nms` (CPU kernel) defines it.
 * mmcv-full (pinned 1.3.2 <= mmcv <= 1.4.0 by /root/reference/mmdet/__init__.py:18-26) is
 * NOT vendored in the reference and is absent from this image, so this restates its
 * published algorithm; the reference's call site is
 * mmdet/core/post_processing/bbox_nms.py:84 (batched_nms -> nms).  "Parity unpinned"
 * against mmcv itself: no reference test holds a known answer for it.
 *
 * Definition (the one the HIP kernel is held to, bit for bit):
 *   areas[i] = (x2 - x1 + offset) * (y2 - y1 + offset),  offset = 0
 *   order    = scores descending; ties broken by ASCENDING index (mmcv uses an unstable
 *              sort, so ties are unspecified there)
 *   for each not-yet-suppressed i in order, for each later j in order:
 *       w = max(0, min(x2i,x2j) - max(x1i,x1j));  h likewise;  inter = w*h
 *       ovr = inter / (areas[i] + areas[j] - inter)          (fp32, IEEE division)
 *       if (ovr > iou_threshold) suppress j
 * Compiled with -ffp-contract=off: no fused multiply-add may change a rounding.
 */
#include <stdint.h>
#include <stdlib.h>

typedef struct { float score; int64_t idx; } ord_t;

static int cmp_ord(const void* a, const void* b) {
  const ord_t* x = (const ord_t*)a;
  const ord_t* y = (const ord_t*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* boxes: n x 4 (x1,y1,x2,y2); keep: room for n indices.  Returns the number kept.
 * form 0: the definition above (mmcv's CPU kernel, IEEE division);
 * form 1: `inter > iou_threshold * (areas[i] + areas[j] - inter)` -- the predicate of mmcv-full 1.3.x's CUDA kernel
 *         (devIoU in nms_cuda_kernel.cuh), i.e. what the reference runs on a GPU.  Same sort, same greedy loop; the two
 *         forms select differently only where the fp32 quotient / product rounds across the threshold. */
int64_t oracle_nms_form(const float* boxes, const float* scores, int64_t n, float iou_threshold, int64_t* keep, int form);

int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float iou_threshold, int64_t* keep) {
  return oracle_nms_form(boxes, scores, n, iou_threshold, keep, 0);
}

int64_t oracle_nms_form(const float* boxes, const float* scores, int64_t n, float iou_threshold, int64_t* keep, int form) {
  if (n <= 0) return 0;
  ord_t* order = (ord_t*)malloc((size_t)n * sizeof(ord_t));
  float* areas = (float*)malloc((size_t)n * sizeof(float));
  unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
  for (int64_t i = 0; i < n; ++i) {
    order[i].score = scores[i];
    order[i].idx = i;
    areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
  }
  qsort(order, (size_t)n, sizeof(ord_t), cmp_ord);
  int64_t k = 0;
  for (int64_t _i = 0; _i < n; ++_i) {
    if (dead[_i]) continue;
    const int64_t i = order[_i].idx;
    keep[k++] = i;
    const float ix1 = boxes[4 * i + 0], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    const float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < n; ++_j) {
      if (dead[_j]) continue;
      const int64_t j = order[_j].idx;
      const float xx1 = ix1 > boxes[4 * j + 0] ? ix1 : boxes[4 * j + 0];
      const float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
      const float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
      const float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
      float w = xx2 - xx1;
      float h = yy2 - yy1;
      if (!(w > 0.f)) w = 0.f;
      if (!(h > 0.f)) h = 0.f;
      const float inter = w * h;
      const float uni = iarea + areas[j] - inter;
      if (form) {
        const float rhs = iou_threshold * uni;
        if (inter > rhs) dead[_j] = 1;
      } else {
        const float ovr = inter / uni;
        if (ovr > iou_threshold) dead[_j] = 1;
      }
    }
  }
  free(order);
  free(areas);
  free(dead);
  return k;
}
