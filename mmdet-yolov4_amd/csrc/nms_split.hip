// mmcv batched_nms, n >= split_thr branch, for ONE image on gfx950:
//   for id in unique(idxs): keep |= nms(boxes_for_nms[idxs == id], scores[idxs == id])
//   keep = keep.nonzero();  keep = keep[scores[keep].argsort(descending=True)]
// (mmcv-full 1.3.x, third party; call site mmdet/core/post_processing/bbox_nms.py:84).
// Ties in the final argsort are broken by ascending candidate index (mmcv leaves them
// unspecified), the same rule as everywhere else in this library.
//
// Pipeline (all on the caller's stream, n is a host value):
//   1. radix sort of the 64-bit candidate keys (score desc, index asc)           [rs_* kernels below]
//   2. stable radix sort of those by class label -> class-major, order kept      [rs_* kernels below]
//   3. one workgroup per class: greedy NMS over its segment in chunks of 256, survivors
//      of earlier chunks kept as class-offset boxes in a global scratch list
//   4. survivors' keys (others = ~0) sorted again, the first max_out become detections
// Built with -ffp-contract=off (see nms_common.h).
#include "nms_common.h"

namespace yv4 {

constexpr int kSplitThreads = 1024;
constexpr int kSplitChunk = 256;

// ---- stable LSD radix sort, 8 bits per pass (this path is cold: >= 10 000 candidates of one image; three launches per
// pass, nothing tuned).  Per pass: (a) every workgroup counts the digits of its tile of 1 024 keys -> hist[digit][tile];
// (b) one workgroup turns the digit-major table into exclusive offsets; (c) every workgroup scatters its tile, a key's
// position = offset[digit][tile] + its rank among the tile's earlier keys with the same digit.  A workgroup is ONE wave:
// tile order = (round, lane), so a rank is the running count of the digit over earlier rounds plus the number of lower
// lanes with the same digit in this round (eight ballots) -- no cross-wave ordering to get wrong.
constexpr int kRsLanes = 64;
constexpr int kRsRounds = 16;
constexpr int kRsTile = kRsLanes * kRsRounds;

template <class K>
__global__ __launch_bounds__(kRsLanes) void rs_hist_kernel(const K* __restrict__ keys, int64_t n, int shift,
                                                          uint32_t* __restrict__ hist, int ntiles) {
  __shared__ uint32_t cnt[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += kRsLanes) cnt[i] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll 4
  for (int r = 0; r < kRsRounds; ++r) {
    const int64_t idx = base + r * kRsLanes + lane;
    if (idx < n) atomicAdd(&cnt[(unsigned)(keys[idx] >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int i = lane; i < 256; i += kRsLanes) hist[(int64_t)i * ntiles + blockIdx.x] = cnt[i];
}

// exclusive scan of `total` counters in place (their sum is n < 2^31): a thread sums its contiguous chunk, thread 0 scans
// the 1 024 chunk sums, the thread walks its chunk again
__global__ __launch_bounds__(1024) void rs_scan_kernel(uint32_t* __restrict__ h, int64_t total) {
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (total + 1023) / 1024;
  const int64_t lo = t * chunk < total ? t * chunk : total;
  const int64_t hi = lo + chunk < total ? lo + chunk : total;
  uint32_t su = 0u;
  for (int64_t i = lo; i < hi; ++i) su += h[i];
  part[t] = su;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0u;
    for (int i = 0; i < 1024; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
  }
  __syncthreads();
  uint32_t run = part[t];
  for (int64_t i = lo; i < hi; ++i) { const uint32_t v = h[i]; h[i] = run; run += v; }
}

template <class K, class V, bool HAS_V>
__global__ __launch_bounds__(kRsLanes) void rs_scatter_kernel(const K* __restrict__ kin, K* __restrict__ kout,
                                                             const V* __restrict__ vin, V* __restrict__ vout, int64_t n,
                                                             int shift, const uint32_t* __restrict__ offs, int ntiles) {
  __shared__ uint32_t run[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += kRsLanes) run[i] = offs[(int64_t)i * ntiles + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kRsRounds; ++r) {           // (uniform trip count: the barriers below are reached by every lane)
    const int64_t idx = base + r * kRsLanes + lane;
    const bool valid = idx < n;
    const K k = valid ? kin[idx] : (K)0;
    const unsigned d = (unsigned)(k >> shift) & 255u;
    unsigned long long peers = __ballot(valid);    // lanes of this round with my digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const unsigned rank = (unsigned)__popcll(peers & below);
    if (valid) {
      const uint32_t pos = run[d] + rank;
      kout[pos] = k;
      if (HAS_V) vout[pos] = vin[idx];
    }
    __syncthreads();                               // every read of run[] of this round is done
    if (valid && rank + 1u == (unsigned)__popcll(peers)) run[d] += (uint32_t)__popcll(peers);   // the group's highest lane
    __syncthreads();
  }
}

// Sorts `bits` low bits (a multiple of 16: an even number of passes) of n keys, ascending and stable; values follow when
// HAS_V.  The input is only read; the result lands in (kx, vx), (ky, vy) is the other side of the ping-pong.
template <class K, class V, bool HAS_V>
static int rs_sort(const K* kin, K* kx, K* ky, const V* vin, V* vx, V* vy, int64_t n, int bits, uint32_t* hist, hipStream_t s) {
  const int ntiles = (int)((n + kRsTile - 1) / kRsTile);
  const K* sk = kin;
  const V* sv = vin;
  for (int pass = 0; pass * 8 < bits; ++pass) {
    K* dk = (pass & 1) ? kx : ky;
    V* dv = (pass & 1) ? vx : vy;
    hipLaunchKernelGGL(rs_hist_kernel<K>, dim3((unsigned)ntiles), dim3(kRsLanes), 0, s, sk, n, pass * 8, hist, ntiles);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, hist, (int64_t)256 * ntiles);
    hipLaunchKernelGGL((rs_scatter_kernel<K, V, HAS_V>), dim3((unsigned)ntiles), dim3(kRsLanes), 0, s, sk, dk, sv, dv, n,
                       pass * 8, hist, ntiles);
    sk = dk;
    sv = dv;
  }
  YV4_CHECK_LAUNCH("nms_split: radix sort");
  return YV4_OK;
}

__global__ __launch_bounds__(256) void split_labels_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                           const int32_t* __restrict__ labels, int fused,
                                                           int32_t* __restrict__ out_labels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t flat = (uint32_t)keys[i];
  out_labels[i] = fused > 0 ? (int32_t)(flat % (uint32_t)fused) : (labels ? labels[flat] : 0);
}

// seg[c] = first sorted position with label >= c  (seg has num_classes + 1 entries)
__global__ __launch_bounds__(256) void split_segments_kernel(const int32_t* __restrict__ sorted_labels, int64_t n,
                                                             int num_classes, int64_t* __restrict__ seg) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > num_classes) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_labels[mid] < c) lo = mid + 1; else hi = mid;
  }
  seg[c] = lo;
}

struct SplitArgs {
  const uint64_t* keys;      // class-major sorted candidate keys
  const int64_t* seg;        // num_classes + 1 segment bounds
  const float* boxes;
  int fused;
  float off_unit;            // max_coord + 1
  float iou_thr;
  int iou_form;
  float4* kept_box;          // scratch, n entries (class segment c uses [seg[c], ...))
  float* kept_area;          // scratch, n entries
  uint64_t* out_keys;        // n entries: key if kept else ~0
};

__global__ __launch_bounds__(kSplitThreads) void split_class_nms_kernel(SplitArgs p) {
  __shared__ float4 cbox[kSplitChunk];
  __shared__ float carea[kSplitChunk];
  __shared__ uint64_t cmask[kSplitChunk * 4];
  __shared__ uint64_t calive[4];
  __shared__ int kcount;
  const int cls = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t lo = p.seg[cls], hi = p.seg[cls + 1];
  const int64_t n = hi - lo;
  if (n <= 0) return;
  const uint64_t* keys = p.keys + lo;
  float4* kbox = p.kept_box + lo;
  float* karea = p.kept_area + lo;
  uint64_t* okeys = p.out_keys + lo;
  const float off = (float)cls * p.off_unit;
  if (tid == 0) kcount = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < n; c0 += kSplitChunk) {
    const int cn = (int)min((int64_t)kSplitChunk, n - c0);
    const int kept = kcount;
    if (tid < kSplitChunk) {
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      float ar = 0.f;
      if (tid < cn) {
        const uint32_t flat = (uint32_t)keys[c0 + tid];
        const uint32_t bi = p.fused > 0 ? flat / (uint32_t)p.fused : flat;
        const float4 ob = reinterpret_cast<const float4*>(p.boxes)[bi];
        bb = make_float4(ob.x + off, ob.y + off, ob.z + off, ob.w + off);
        ar = (bb.z - bb.x) * (bb.w - bb.y);
      }
      cbox[tid] = bb;
      carea[tid] = ar;
    }
    __syncthreads();
    {  // (a) chunk vs survivors of earlier chunks (global scratch, written by this workgroup)
      const int i = tid & (kSplitChunk - 1);
      const int q = tid >> 8;
      bool dead = i >= cn;
      if (!dead) {
        const float4 bj = cbox[i];
        const float aj = carea[i];
        const volatile float4* vb = kbox;
        const volatile float* va = karea;
        for (int k = q; k < kept && !dead; k += 4) {
          float4 bk;
          bk.x = vb[k].x; bk.y = vb[k].y; bk.z = vb[k].z; bk.w = vb[k].w;
          dead = iou_gt(bk, va[k], bj, aj, p.iou_thr, p.iou_form);
        }
      }
      const unsigned long long live = __ballot(!dead);
      if (q == 0 && (tid & 63) == 0) calive[tid >> 6] = live;
      __syncthreads();
      if (q != 0 && (tid & 63) == 0) atomicAnd(reinterpret_cast<unsigned long long*>(&calive[(tid & 255) >> 6]), live);
    }
    {  // (b) chunk x chunk bitmask
      const int i = tid >> 2;
      const int w = tid & 3;
      uint64_t bits = 0;
      if (i < cn) {
        const float4 bi = cbox[i];
        const float ai = carea[i];
        for (int jj = 0; jj < 64; ++jj) {
          const int j = w * 64 + jj;
          if (j > i && j < cn && iou_gt(bi, ai, cbox[j], carea[j], p.iou_thr, p.iou_form)) bits |= 1ull << jj;
        }
      }
      cmask[i * 4 + w] = bits;
    }
    __syncthreads();
    if (tid == 0) {  // (c) greedy resolve
      uint64_t removed[4] = {0, 0, 0, 0};
      uint64_t keepbits[4] = {0, 0, 0, 0};
      int k = kept;
      for (int w = 0; w < 4; ++w) {
        uint64_t cur = calive[w] & ~removed[w];
        while (cur) {
          const int b = __builtin_ctzll(cur);
          const int i = w * 64 + b;
          kbox[k] = cbox[i];
          karea[k] = carea[i];
          ++k;
          keepbits[w] |= 1ull << b;
          removed[0] |= cmask[i * 4 + 0];
          removed[1] |= cmask[i * 4 + 1];
          removed[2] |= cmask[i * 4 + 2];
          removed[3] |= cmask[i * 4 + 3];
          const uint64_t above = b == 63 ? 0ull : (~0ull << (b + 1));
          cur = calive[w] & ~removed[w] & above;
        }
      }
      kcount = k;
      calive[0] = keepbits[0]; calive[1] = keepbits[1]; calive[2] = keepbits[2]; calive[3] = keepbits[3];
    }
    __threadfence_block();
    __syncthreads();
    if (tid < cn) okeys[c0 + tid] = ((calive[tid >> 6] >> (tid & 63)) & 1ull) ? keys[c0 + tid] : ~0ull;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void split_emit_kernel(const uint64_t* __restrict__ sorted_keys, int64_t n,
                                                         const float* __restrict__ boxes,
                                                         const int32_t* __restrict__ labels, int fused, int max_out,
                                                         float* out_dets, int32_t* out_labels, int64_t* out_index,
                                                         int32_t* out_count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int lim = (int)min((int64_t)max_out, n);
  if (k < lim) {
    const uint64_t key = sorted_keys[k];
    if (key != ~0ull) {
      const uint32_t flat = (uint32_t)key;
      const uint32_t bi = fused > 0 ? flat / (uint32_t)fused : flat;
      const float4 ob = reinterpret_cast<const float4*>(boxes)[bi];
      out_dets[k * 5 + 0] = ob.x; out_dets[k * 5 + 1] = ob.y; out_dets[k * 5 + 2] = ob.z; out_dets[k * 5 + 3] = ob.w;
      out_dets[k * 5 + 4] = key_to_score((uint32_t)(key >> 32));
      out_labels[k] = fused > 0 ? (int32_t)(flat % (uint32_t)fused) : (labels ? labels[flat] : 0);
      out_index[k] = (int64_t)flat;
    }
  }
  if (k == 0) {  // count = number of valid keys among the first lim (valid keys sort first)
    int64_t lo = 0, hi = lim;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_keys[mid] != ~0ull) lo = mid + 1; else hi = mid;
    }
    *out_count = (int32_t)lo;
  }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SplitLayout {
  size_t keys_a, keys_b, keys_t, lab_a, lab_b, lab_t, hist, seg, kbox, karea, total;
};

static SplitLayout split_layout(int64_t n, int num_classes) {
  SplitLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  L.keys_a = take((size_t)n * 8);
  L.keys_b = take((size_t)n * 8);
  L.lab_a = take((size_t)n * 4);
  L.lab_b = take((size_t)n * 4);
  L.seg = take((size_t)(num_classes + 2) * 8);
  L.kbox = take((size_t)n * 16);
  L.karea = take((size_t)n * 4);
  L.keys_t = take((size_t)n * 8);                                              // the sorts' other ping-pong side
  L.lab_t = take((size_t)n * 4);
  L.hist = take((size_t)256 * (size_t)((n + kRsTile - 1) / kRsTile) * 4);     // digit-major counters of a pass
  L.total = off;
  return L;
}

constexpr int kSplitMaxClasses = 65535;

}  // namespace yv4

using namespace yv4;

extern "C" size_t yv4_nms_split_work(int64_t n) {
  if (n <= 0 || n >= (1LL << 31)) return 0;
  return split_layout(n, kSplitMaxClasses).total;
}

extern "C" int yv4_nms_split(const uint64_t* keys, int64_t n, float max_coord, const float* boxes,
                             const int32_t* labels, int fused_classes, float iou_thr, int max_out, void* work,
                             float* out_dets, int32_t* out_labels, int64_t* out_index, int32_t* out_count,
                             void* stream) {
  YV4_REQUIRE(keys && boxes && work && out_dets && out_labels && out_index && out_count, "nms_split: null pointer");
  YV4_REQUIRE(n > 0 && n < (1LL << 31), "nms_split: n out of range");
  YV4_REQUIRE(max_out > 0 && fused_classes >= 0 && fused_classes <= kSplitMaxClasses, "nms_split: bad max_out / classes");
  YV4_REQUIRE(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)work & 255) == 0, "nms_split: boxes must be 16-byte and work 256-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // labels are < 65536 (16 radix bits); with fused classes the class count is known, otherwise
  // the caller's labels are bounded by the same limit
  const int num_classes = fused_classes > 0 ? fused_classes : kSplitMaxClasses;
  const SplitLayout L = split_layout(n, kSplitMaxClasses);
  char* w = reinterpret_cast<char*>(work);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(w + L.keys_a);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(w + L.keys_b);
  int32_t* lab_a = reinterpret_cast<int32_t*>(w + L.lab_a);
  int32_t* lab_b = reinterpret_cast<int32_t*>(w + L.lab_b);
  int64_t* seg = reinterpret_cast<int64_t*>(w + L.seg);
  uint64_t* keys_t = reinterpret_cast<uint64_t*>(w + L.keys_t);
  uint32_t* lab_t = reinterpret_cast<uint32_t*>(w + L.lab_t);
  uint32_t* hist = reinterpret_cast<uint32_t*>(w + L.hist);
  const unsigned g = (unsigned)((n + 255) / 256);
  // 1. by (score desc, index asc)
  if (int rc = rs_sort<uint64_t, int, false>(keys, keys_a, keys_t, nullptr, nullptr, nullptr, n, 64, hist, s)) return rc;
  // 2. stable by label (labels are < 65536: two passes)
  hipLaunchKernelGGL(split_labels_kernel, dim3(g), dim3(256), 0, s, keys_a, n, labels, fused_classes, lab_a);
  if (int rc = rs_sort<uint32_t, uint64_t, true>(reinterpret_cast<const uint32_t*>(lab_a), reinterpret_cast<uint32_t*>(lab_b),
                                                 lab_t, keys_a, keys_b, keys_t, n, 16, hist, s))
    return rc;
  hipLaunchKernelGGL(split_segments_kernel, dim3((num_classes + 1 + 255) / 256), dim3(256), 0, s, lab_b, n, num_classes, seg);
  // 3. per-class NMS; survivors' keys into keys_a (others ~0)
  SplitArgs a;
  a.keys = keys_b; a.seg = seg; a.boxes = boxes; a.fused = fused_classes; a.off_unit = max_coord + 1.f;
  a.iou_thr = iou_thr; a.iou_form = nms_iou_form(); a.kept_box = reinterpret_cast<float4*>(w + L.kbox); a.kept_area = reinterpret_cast<float*>(w + L.karea);
  a.out_keys = keys_a;
  hipLaunchKernelGGL(split_class_nms_kernel, dim3(num_classes), dim3(kSplitThreads), 0, s, a);
  // 4. survivors by (score desc, index asc), first max_out
  if (int rc = rs_sort<uint64_t, int, false>(keys_a, keys_b, keys_t, nullptr, nullptr, nullptr, n, 64, hist, s)) return rc;
  const int lim = (int)(n < max_out ? n : max_out);
  hipLaunchKernelGGL(split_emit_kernel, dim3((lim + 255) / 256), dim3(256), 0, s, keys_b, n, boxes, labels, fused_classes,
                     max_out, out_dets, out_labels, out_index, out_count);
  YV4_CHECK_LAUNCH("nms_split");
  return YV4_OK;
}
