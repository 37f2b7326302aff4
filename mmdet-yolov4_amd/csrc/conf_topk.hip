// nms_pre: top-k of the objectness as an ADMISSION THRESHOLD, by radix select (no sort, no library).
//
//   yolocsp_head.py:349-355   `_, topk_inds = conf_pred.topk(nms_pre)` per image over all levels      -> yv4_conf_topk
//   yolo_head.py:281-303      the same inside the level loop of YOLOV3Head, only on levels with more than nms_pre boxes
//                             (core/export/onnx_helper.py:45-78)                                       -> yv4_conf_topk_levels
//
// One key per anchor box, (order(conf) << 32 | box index): ascending key = descending conf, ties by ascending box index
// (torch.topk leaves ties unspecified).  Keys are unique, so "the k-th smallest key of a segment" is well defined and the
// decode kernel admits exactly the boxes whose key is <= it -- the index list of topk is never materialised.  Round 2
// obtained that key by sorting every segment (hipcub::DeviceSegmentedRadixSort over N x 22 743 keys per YOLOv3 step); only
// ONE order statistic is needed: a most-significant-byte-first radix SELECT, one workgroup per segment, eight passes of
// a 256-bin histogram in LDS over the segment's keys (182 KB per image, L2-resident).  Pass p keeps the items whose top p
// bytes equal the prefix found so far, counts them by their next byte, and the bin in which the running count crosses the
// remaining rank extends the prefix.  After the fourth pass the candidates are the boxes whose conf equals the k-th conf
// bit for bit (normally one), and the last four passes rank those by index.
// Built with -ffp-contract=off (see nms_common.h): the sigmoid is the decode kernel's.
#include "nms_common.h"

namespace yv4 {

struct ConfKeyArgs {
  const float* pred[8];
  int boxes[8];        // H*W*A per level
  int level_base[9];
  int num_levels, attr, total;
};

__global__ __launch_bounds__(256) void conf_keys_kernel(ConfKeyArgs p, uint64_t* __restrict__ keys) {
  const int n = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.total) return;
  int lvl = 0;
  while (lvl + 1 < p.num_levels && j >= p.level_base[lvl + 1]) ++lvl;
  const int jl = j - p.level_base[lvl];
  const float logit = p.pred[lvl][((size_t)n * p.boxes[lvl] + jl) * p.attr + 4];
  keys[(size_t)n * p.total + j] = ((uint64_t)score_to_key(sigmoid_f32(logit)) << 32) | (uint32_t)j;
}

constexpr int kSelThreads = 1024;

// Segment s of the launch: keys[seg_begin(s) .. seg_begin(s) + seg_len(s)); LEVELS = false: one segment per image (all
// levels), LEVELS = true: one per (image, level).  out[s] = the k-th smallest key, or ~0 (admit everything) when the
// segment has no more than k keys.
template <bool LEVELS>
__global__ __launch_bounds__(kSelThreads) void conf_select_kernel(const uint64_t* __restrict__ keys, ConfKeyArgs a, int k,
                                                                  uint64_t* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_rank;
  const int seg = blockIdx.x;
  int n_img, begin, len;
  if (LEVELS) {
    n_img = seg / a.num_levels;
    const int l = seg - n_img * a.num_levels;
    begin = a.level_base[l];
    len = a.boxes[l];
  } else {
    n_img = seg;
    begin = 0;
    len = a.total;
  }
  if (len <= k) {                                   // (block-uniform) nothing to cut on this segment
    if (threadIdx.x == 0) out[seg] = ~0ull;
    return;
  }
  const uint64_t* kp = keys + (size_t)n_img * a.total + begin;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) { s_prefix = 0ull; s_rank = (unsigned)k; }       // rank is 1-based: the k-th smallest
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    for (int i = tid; i < ((len + kSelThreads - 1) / kSelThreads) * kSelThreads; i += kSelThreads) {
      bool live = false;
      unsigned bin = 0u;
      if (i < len) {
        const unsigned long long key = kp[i];
        live = pass == 0 || (key >> (shift + 8)) == prefix;
        bin = (unsigned)(key >> shift) & 255u;
      }
      // objectness of an image clusters in a few exponent bins: when every live lane of the wave hits the same bin one
      // lane adds the count (64 same-address LDS atomics would serialise), otherwise plain atomics
      const unsigned long long m = __ballot(live);
      if (m) {
        const unsigned b0 = __builtin_amdgcn_readfirstlane(__shfl(bin, __ffsll((long long)m) - 1));
        const unsigned long long same = __ballot(live && bin == b0);
        if (same == m) {
          if (lane == __ffsll((long long)m) - 1) atomicAdd(&hist[b0], (unsigned)__popcll(m));
        } else if (live) {
          atomicAdd(&hist[bin], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {                                 // one wave scans the 256 bins: 4 per lane, prefix sum across lanes
      const unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
      const unsigned mine = c0 + c1 + c2 + c3;
      unsigned incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      const unsigned before = incl - mine;
      const unsigned rank = s_rank;
      if (before < rank && rank <= incl) {          // exactly one lane: the crossing lies in its four bins
        unsigned rem = rank - before;
        unsigned b = 4u * lane;
        if (rem > c0) { rem -= c0; ++b; if (rem > c1) { rem -= c1; ++b; if (rem > c2) { rem -= c2; ++b; } } }
        s_prefix = (prefix << 8) | b;
        s_rank = rem;
      }
    }
    __syncthreads();
  }
  if (tid == 0) out[seg] = s_prefix;
}

static size_t topk_key_bytes(int N, long long total) {
  return ((size_t)N * (size_t)total * sizeof(uint64_t) + 255) / 256 * 256;
}

static int fill_key_args(const yv4_level_desc* levels, int num_levels, int A, int num_classes, ConfKeyArgs& a, const char* who) {
  long long total = 0;
  for (int l = 0; l < num_levels; ++l) {
    YV4_REQUIRE(levels[l].pred && levels[l].H > 0 && levels[l].W > 0, "%s: level %d is malformed", who, l);
    a.pred[l] = levels[l].pred;
    a.boxes[l] = levels[l].H * levels[l].W * A;
    a.level_base[l] = (int)total;
    total += a.boxes[l];
  }
  a.level_base[num_levels] = (int)total;
  a.num_levels = num_levels; a.attr = 5 + num_classes; a.total = (int)total;
  return YV4_OK;
}

}  // namespace yv4

using namespace yv4;

extern "C" size_t yv4_conf_topk_levels_work(int N, int64_t total_anchors, int num_levels) {
  if (N <= 0 || total_anchors <= 0 || num_levels <= 0 || (long long)N * total_anchors >= (1LL << 31)) return 0;
  return topk_key_bytes(N, total_anchors);
}

extern "C" size_t yv4_conf_topk_work(int N, int64_t total_anchors) {
  if (N <= 0 || total_anchors <= 0 || (long long)N * total_anchors >= (1LL << 31)) return 0;
  return topk_key_bytes(N, total_anchors);
}

extern "C" int yv4_conf_topk_levels(const yv4_level_desc* levels, int num_levels, int N, int A, int num_classes, int k,
                                    void* work, uint64_t* topk_keys, void* stream) {
  YV4_REQUIRE(levels && work && topk_keys, "conf_topk_levels: null pointer");
  YV4_REQUIRE(num_levels > 0 && num_levels <= 8 && N > 0 && A > 0 && A <= 8 && num_classes >= 0 && k > 0,
              "conf_topk_levels: bad sizes");
  ConfKeyArgs a;
  if (int rc = fill_key_args(levels, num_levels, A, num_classes, a, "conf_topk_levels")) return rc;
  YV4_REQUIRE((long long)N * a.total < (1LL << 31), "conf_topk_levels: N * anchors overflows int32");
  uint64_t* keys = static_cast<uint64_t*>(work);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(conf_keys_kernel, dim3((unsigned)((a.total + 255) / 256), N), dim3(256), 0, s, a, keys);
  hipLaunchKernelGGL(conf_select_kernel<true>, dim3((unsigned)(N * num_levels)), dim3(kSelThreads), 0, s, keys, a, k, topk_keys);
  YV4_CHECK_LAUNCH("conf_topk_levels");
  return YV4_OK;
}

extern "C" int yv4_conf_topk(const yv4_level_desc* levels, int num_levels, int N, int A, int num_classes, int k, void* work,
                             uint64_t* topk_keys, void* stream) {
  YV4_REQUIRE(levels && work && topk_keys, "conf_topk: null pointer");
  YV4_REQUIRE(num_levels > 0 && num_levels <= 8 && N > 0 && A > 0 && A <= 8 && num_classes >= 0, "conf_topk: bad sizes");
  ConfKeyArgs a;
  if (int rc = fill_key_args(levels, num_levels, A, num_classes, a, "conf_topk")) return rc;
  YV4_REQUIRE(k > 0 && k < a.total, "conf_topk: need 0 < k < anchors per image (k = %d, anchors = %d)", k, a.total);
  YV4_REQUIRE((long long)N * a.total < (1LL << 31), "conf_topk: N * anchors overflows int32");
  uint64_t* keys = static_cast<uint64_t*>(work);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(conf_keys_kernel, dim3((unsigned)((a.total + 255) / 256), N), dim3(256), 0, s, a, keys);
  hipLaunchKernelGGL(conf_select_kernel<false>, dim3((unsigned)N), dim3(kSelThreads), 0, s, keys, a, k, topk_keys);
  YV4_CHECK_LAUNCH("conf_topk");
  return YV4_OK;
}
