// Post-processing of the YOLOv4 head on gfx950: sigmoid + anchor decode + score
// threshold (one pass over the NHWC pred maps) and per-image batched NMS.
//
// Built with -ffp-contract=off: the reference computes these in separate fp32 ops
// (mul then add, never fma) and the NMS decision `inter/(a+b-inter) > thr` must be
// reproduced bit for bit.
//
// Reference code restated here:
//   decode   mmdet/models/dense_heads/yolocsp_head.py:263-285 (sigmoid, 2s-1, (2s)^2),
//            mmdet/core/bbox/coder/yolov4_bbox_coder.py:51-67,
//            mmdet/core/anchor/anchor_generator.py:255-270 (grid = base + shift),
//            yolocsp_head.py:357-366 (cls *= conf, boxes /= scale_factor),
//            mmdet/core/post_processing/bbox_nms.py:54,66 (scores > score_thr)
//   nms      mmcv.ops.nms.batched_nms / nms (mmcv-full 1.3.x, not vendored in the
//            reference; call site bbox_nms.py:84): class offset box + label*(max+1),
//            order by score descending, greedy suppression when
//            inter / (area_i + area_j - inter) > iou_threshold, offset = 0.
//            Tie-break (unspecified by mmcv): lower flat candidate index first.
#include "yv4_common.h"
#include "nms_common.h"

namespace yv4 {

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  // valid for any mix of signs when *addr starts at -inf
  if (v >= 0.f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void decode_reset_kernel(int32_t* counts, float* max_coord, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    counts[i] = 0;
    max_coord[i] = -__builtin_huge_valf();
  }
}

// ---------------------------------------------------------------------------------
// decode + filter.  A workgroup owns 64 consecutive anchor boxes of one (image, level):
// their 64*(5+C) logits are contiguous in the NHWC pred map, are read once, coalesced,
// and go through sigmoid into LDS.  Then thread t works on box t/4 and classes
// t%4, t%4+4, ...
// ---------------------------------------------------------------------------------
constexpr int kDecBoxes = 64;
constexpr int kMaxLevels = 8;

struct DecodeArgs {
  const float* pred[kMaxLevels];
  int H[kMaxLevels], W[kMaxLevels], stride[kMaxLevels];
  int level_base[kMaxLevels];   // first anchor-box index of the level inside an image
  int block_base[kMaxLevels + 1];  // first workgroup (per image) of the level
  float base[kMaxLevels][8][4];
  int num_levels, N, A, C, total_anchors;
  float score_thr;
  const float* scale_factor;
  float* boxes;
  float* conf;
  float* cls;
  uint64_t* keys;
  int64_t key_cap;
  int32_t* counts;
  float* max_coord;
  const uint64_t* topk;   // largest admissible (conf key << 32 | anchor) per image (or per image x level), or NULL
  int v3;                 // YOLOV3Head semantics (yolo_head.py:254-391) instead of YOLOCSPHead's
  int topk_per_level;     // topk has N * num_levels entries (YOLOv3 selects the top-k per level)
  float conf_thr;         // v3: boxes with objectness < conf_thr are dropped (<= 0: off)
  int ablate;             // measurement only (YV4_DEC_ABLATE): 1 no sigmoid, 2 no candidate output, 4 no class loop, 8 no box stores
};

// A workgroup walks kDecTiles consecutive 64-box tiles of one image.  Candidates collect in an LDS key buffer and leave with
// ONE reservation on the image's counter per workgroup (a returning device-scope atomic per 64 boxes -- 11 400 per step on
// the 32 words of one cache line at batch 32 -- was most of what the kernel did beyond reading its logits); a tile with
// more candidates than the buffer holds reserves and writes directly, a full buffer is flushed early.  Two tiles: 121 ->
// 104 us at batch 32; four 108, eight 128 (the tiles of a workgroup are serial round trips).  Tried and dropped: raw logits
// in the tile and sigmoids only where a value is used (a box whose conf fails the threshold has no candidate) -- 130 us:
// on the benchmark's pred maps most boxes pass on conf, and their class scores are then computed twice.
constexpr int kDecTiles = 2;
constexpr int kDecKeyBuf = 1024;

__global__ __launch_bounds__(256) void decode_filter_kernel(DecodeArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [64][attr] sigmoid values
  __shared__ uint64_t kbuf[kDecKeyBuf];
  __shared__ int wg_count, wg_base, kcnt;
  const int n = blockIdx.y;
  const int attr = 5 + p.C;
  const int nblocks = p.block_base[p.num_levels];
  uint64_t* const ikeys = p.keys + (size_t)n * p.key_cap;
  if (threadIdx.x == 0) kcnt = 0;
  // the buffered keys leave: one reservation, coalesced stores (uniform: every thread calls it)
  auto flush = [&]() {
    const int cnt = kcnt;
    if (cnt > 0) {
      if (threadIdx.x == 0) wg_base = atomicAdd(&p.counts[n], cnt);
      __syncthreads();
      const int base = wg_base;
      for (int i = threadIdx.x; i < cnt; i += 256)
        if (base + i < p.key_cap) ikeys[base + i] = kbuf[i];
      __syncthreads();
      if (threadIdx.x == 0) kcnt = 0;
      __syncthreads();
    }
  };
  float mx_acc = -__builtin_huge_valf();
  for (int tl = 0; tl < kDecTiles; ++tl) {
  const int bx = (int)blockIdx.x * kDecTiles + tl;
  if (bx >= nblocks) break;
  int lvl = 0;
  while (lvl + 1 < p.num_levels && bx >= p.block_base[lvl + 1]) ++lvl;
  const int boxes_lvl = p.H[lvl] * p.W[lvl] * p.A;
  const int b0 = (bx - p.block_base[lvl]) * kDecBoxes;  // first box of this tile in the level
  const int nb = min(kDecBoxes, boxes_lvl - b0);
  const float* src = p.pred[lvl] + ((size_t)n * boxes_lvl + b0) * attr;
  const int nval = nb * attr;
  if (p.v3) {
    for (int i = threadIdx.x; i < nval; i += 256) {
      const int at = i % attr;
      sm[i] = (at == 2 || at == 3) ? src[i] : sigmoid_f32(src[i]);   // v3: exp(t_w), exp(t_h) need the raw logit
    }
  } else if ((((uintptr_t)src) & 15) == 0) {
    // every attribute goes through the sigmoid: no per-element index work, and the workgroup's values (64 x 85 floats,
    // a multiple of 16 bytes from an aligned start for all but odd images of an odd-sized level) are read as 16-byte
    // words, all of a thread's loads issued before the first sigmoid -- the scalar loop was a chain of ~21 load ->
    // sigmoid -> LDS-store rounds per thread and set the kernel's time (one memory round trip each)
    const int nq = nval >> 2;                                // whole 16-byte words; nval % 4 != 0 only in a level's last block
    constexpr int kQ = (kDecBoxes * (5 + 80) / 4 + 255) / 256;   // 6 words per thread cover 80 classes; more classes: loop
    const float4* src4 = reinterpret_cast<const float4*>(src);
    for (int q0 = threadIdx.x; q0 < nq; q0 += 256 * kQ) {
      float4 v[kQ];
#pragma unroll
      for (int u = 0; u < kQ; ++u) {
        const int q = q0 + 256 * u;
        v[u] = q < nq ? src4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < kQ; ++u) {
        const int q = q0 + 256 * u;
        if (q < nq)
          reinterpret_cast<float4*>(sm)[q] = YV4_ABLATE(p.ablate, 1) ? v[u] : make_float4(sigmoid_f32(v[u].x), sigmoid_f32(v[u].y), sigmoid_f32(v[u].z), sigmoid_f32(v[u].w));
      }
    }
    for (int i = (nq << 2) + threadIdx.x; i < nval; i += 256) sm[i] = sigmoid_f32(src[i]);
  } else {
    for (int i = threadIdx.x; i < nval; i += 256) sm[i] = sigmoid_f32(src[i]);
  }
  if (threadIdx.x == 0) wg_count = 0;
  __syncthreads();

  const int part = threadIdx.x & 3;
  const int b = min((int)(threadIdx.x >> 2), nb - 1);   // threads past the tail recompute the last box, emit nothing
  const bool live = (int)(threadIdx.x >> 2) < nb;
  const float* s = sm + b * attr;
  const int jl = b0 + b;  // box index within the level: (y*W + x)*A + a
  const int a = jl % p.A;
  const int cell = jl / p.A;
  const int gx = cell % p.W[lvl];
  const int gy = cell / p.W[lvl];
  const float stride = (float)p.stride[lvl];
  // anchor = base + shift  (anchor_generator.py:255-266)
  const float sx = (float)(gx * p.stride[lvl]);
  const float sy = (float)(gy * p.stride[lvl]);
  const float ax1 = p.base[lvl][a][0] + sx, ay1 = p.base[lvl][a][1] + sy;
  const float ax2 = p.base[lvl][a][2] + sx, ay2 = p.base[lvl][a][3] + sy;
  const float xc = (ax1 + ax2) * 0.5f, yc = (ay1 + ay2) * 0.5f;
  const float aw = ax2 - ax1, ah = ay2 - ay1;
  float xcp, ycp, wp, hp;
  if (p.v3) {
    // core/bbox/coder/yolo_bbox_coder.py:76-83
    xcp = (s[0] - 0.5f) * stride + xc;
    ycp = (s[1] - 0.5f) * stride + yc;
    wp = expf(s[2]) * aw;
    hp = expf(s[3]) * ah;
  } else {
    // yolocsp_head.py:273-275, yolov4_bbox_coder.py:51-66
    const float px = s[0] * 2.f - 1.f;
    const float py = s[1] * 2.f - 1.f;
    const float tw = s[2] * 2.f, th = s[3] * 2.f;
    xcp = px * stride + xc;
    ycp = py * stride + yc;
    wp = tw * tw * aw;
    hp = th * th * ah;
  }
  float x1 = xcp - wp / 2.f, y1 = ycp - hp / 2.f, x2 = xcp + wp / 2.f, y2 = ycp + hp / 2.f;
  if (p.scale_factor) {  // yolocsp_head.py:365-366
    const float* sf = p.scale_factor + n * 4;
    x1 /= sf[0]; y1 /= sf[1]; x2 /= sf[2]; y2 /= sf[3];
  }
  const float cf = s[4];
  const int j = p.level_base[lvl] + jl;
  const size_t gj = (size_t)n * p.total_anchors + j;
  if (part == 0 && live && !YV4_ABLATE(p.ablate, 8)) {
    reinterpret_cast<float4*>(p.boxes)[gj] = make_float4(x1, y1, x2, y2);
    if (p.conf) p.conf[gj] = cf;
  }
  // nms_pre (yolocsp_head.py:349-355): only the top-k anchors by conf stay candidates; the
  // k-th (conf, anchor) key of the image was selected by yv4_conf_topk
  const uint64_t ckey = ((uint64_t)score_to_key(cf) << 32) | (uint32_t)j;
  const bool in_topk = !p.topk || ckey <= p.topk[p.topk_per_level ? n * p.num_levels + lvl : n];
  // v3: per-image objectness threshold applied after the top-k (yolo_head.py:366-377, `ge`)
  const bool admitted = live && in_topk && !(p.v3 && p.conf_thr > 0.f && !(cf >= p.conf_thr));
  // Pass 1 counts this tile's candidates (one LDS atomic per thread that has any), pass 2 writes the keys -- into the
  // LDS buffer, or straight to the image's key buffer when the tile alone exceeds it.  The order of keys inside an
  // image's buffer is irrelevant (NMS sorts them).
  int mine = 0;
  if (admitted && !YV4_ABLATE(p.ablate, 4)) {
    if (p.C == 0) {                 // class_agnostic (yolocsp_head.py:357-360): one column, score = conf
      mine = (part == 0 && cf > p.score_thr) ? 1 : 0;
    } else {
      for (int c = part; c < p.C; c += 4) {
        const float sc = s[5 + c];
        if (p.cls) p.cls[gj * p.C + c] = sc;
        // YOLOCSPHead: cls * conf > thr (yolocsp_head.py:358, bbox_nms.py:54); YOLOV3Head: cls > thr, the
        // objectness multiplies afterwards as multiclass_nms' score_factors (bbox_nms.py:52-62)
        mine += ((p.v3 ? sc : sc * cf) > p.score_thr) ? 1 : 0;
      }
    }
  }
  if (YV4_ABLATE(p.ablate, 2)) mine = 0;
  int slot = mine ? atomicAdd(&wg_count, mine) : 0;
  __syncthreads();
  const int tc = wg_count;                       // this tile's candidates (uniform)
  const bool direct = tc > kDecKeyBuf;
  if (!direct && kcnt + tc > kDecKeyBuf) flush();
  if (direct) {
    if (threadIdx.x == 0) wg_base = atomicAdd(&p.counts[n], tc);
    __syncthreads();
  }
  const int base = direct ? wg_base : kcnt;
  if (mine) {
    slot += base;
    if (p.C == 0) {
      const uint64_t key = ((uint64_t)score_to_key(cf) << 32) | (uint32_t)j;
      if (!direct) kbuf[slot] = key;
      else if (slot < p.key_cap) ikeys[slot] = key;
    } else {
      for (int c = part; c < p.C; c += 4) {
        const float sc = s[5 + c];
        const float score = sc * cf;
        if ((p.v3 ? sc : score) > p.score_thr) {
          const uint32_t flat = (uint32_t)j * (uint32_t)p.C + (uint32_t)c;
          const uint64_t key = ((uint64_t)score_to_key(score) << 32) | flat;
          if (!direct) kbuf[slot] = key;
          else if (slot < p.key_cap) ikeys[slot] = key;
          ++slot;
        }
      }
    }
    mx_acc = fmaxf(mx_acc, fmaxf(fmaxf(x1, y1), fmaxf(x2, y2)));
  }
  __syncthreads();                               // the tile's values and `kcnt` have been read by everyone
  if (threadIdx.x == 0 && !direct) kcnt = base + tc;
  }  // tiles
  __syncthreads();
  flush();
  // boxes.max() over the surviving candidates (mmcv batched_nms): reduced over the wavefront, then over the workgroup's
  // four waves -- ONE atomic per workgroup.  Device-scope atomics on this 8-XCD part execute at the memory side, and the
  // images' maxima are neighbouring words of one cache line: one atomic per wave with a candidate (~35 000 per step at
  // batch 32) was 55 of the kernel's 175 us.  (Reading the running maximum first, to skip atomics that cannot raise it,
  // is slower still: the coherent read of that line queues behind the same atomics -- 153 us read late, 425 read early.)
  float mx = mx_acc;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  __shared__ float wg_max[4];
  if ((threadIdx.x & 63) == 0) wg_max[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0 && !YV4_ABLATE(p.ablate, 32)) {
    mx = fmaxf(fmaxf(wg_max[0], wg_max[1]), fmaxf(wg_max[2], wg_max[3]));
    if (mx > -__builtin_huge_valf()) atomic_max_float(&p.max_coord[n], mx);
  }
}

// ---------------------------------------------------------------------------------
// NMS, one 1024-thread workgroup per image.
//   1. sort the image's keys in LDS (bitonic, padded to a power of two <= 16384) and
//      write them back in place;
//   2. walk the sorted list in chunks of 256: (a) test the chunk against everything
//      kept so far, (b) build the 256x256 suppression bitmask of the chunk, (c) one
//      thread resolves the chunk greedily and appends survivors to the outputs.
// The greedy result equals the sequential loop of mmcv's nms (see oracle/nms_ref.c).
// ---------------------------------------------------------------------------------
constexpr int kNmsThreads = 1024;
constexpr int kChunk = 256;
constexpr int kSortCap = 16384;   // >= split_thr (10000) rounded up to a power of two
constexpr int kKeptLds = 1024;    // kept boxes cached in LDS; the rest are re-read from the outputs

struct NmsArgs {
  uint64_t* keys;
  int64_t key_cap;
  const int32_t* counts;
  const float* max_coord;
  const float* boxes;
  int64_t boxes_per_image;
  const int32_t* labels;
  int64_t label_stride;
  int fused_classes;
  float iou_thr;
  int iou_form;
  int ablate;     // measurement only (YV4_NMS_ABLATE): 1 stop after the sort, 2 no chunk-vs-kept test, 4 no chunk mask, 8 one chunk only
  int max_out;
  int split_thr;
  float* out_dets;
  int32_t* out_labels;
  int64_t* out_index;
  int32_t* out_count;
};

__global__ __launch_bounds__(kNmsThreads) void nms_images_kernel(NmsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const int n = p.counts[img];
  if (n <= 0) {
    if (tid == 0) p.out_count[img] = 0;
    return;
  }
  if (n >= p.split_thr || n > p.key_cap || n > kSortCap) {
    if (tid == 0) p.out_count[img] = -1;  // caller must use yv4_nms_split
    return;
  }
  uint64_t* gkeys = p.keys + (size_t)img * p.key_cap;

  // ---- 1. sort ------------------------------------------------------------------
  {
    uint64_t* sk = reinterpret_cast<uint64_t*>(lds_raw);
    int P = 1;
    while (P < n) P <<= 1;
    if (P < 2) P = 2;
    for (int i = tid; i < P; i += kNmsThreads) sk[i] = i < n ? gkeys[i] : ~0ull;
    __syncthreads();
    // Bitonic network.  Thread t of a pass handles the pair (i, i + j), i = (t / j) 2j + t mod j: for j <= 64 the 64
    // threads of a wave stay inside their own 128 elements, step after step -- such steps need no workgroup barrier
    // between them (a wave's LDS accesses complete in order), only the steps with j >= 128 exchange between waves.
    // 15 barriers instead of 66 at P = 2048.
    for (int k = 2, lk = 1; k <= P; k <<= 1, ++lk) {
      for (int j = k >> 1, lj = lk - 1; j > 0; j >>= 1, --lj) {
        for (int t = tid; t < (P >> 1); t += kNmsThreads) {
          const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1));      // (t / j) 2j + t mod j, j = 2^lj
          const int ixj = i + j;
          const bool up = (i & k) == 0;
          const uint64_t a = sk[i], b = sk[ixj];
          if ((a > b) == up) {
            sk[i] = b;
            sk[ixj] = a;
          }
        }
        const int next_j = j > 1 ? (j >> 1) : k;          // the step after this one ((2k, k) after (k, 1))
        if (j >= 128 || next_j >= 128) {
          __syncthreads();
        } else {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < n; i += kNmsThreads) gkeys[i] = sk[i];
    __syncthreads();
  }

  if (YV4_ABLATE(p.ablate, 1)) { if (tid == 0) p.out_count[img] = 0; return; }
  // ---- 2. greedy NMS over sorted chunks -------------------------------------------
  // LDS carve (aliases the sort buffer)
  float4* cbox = reinterpret_cast<float4*>(lds_raw);                       // [256] class-offset boxes
  float* carea = reinterpret_cast<float*>(cbox + kChunk);                  // [256]
  uint64_t* cmask = reinterpret_cast<uint64_t*>(carea + kChunk);           // [256][4]
  uint64_t* calive = cmask + kChunk * 4;                                   // [4]
  float4* cobox = reinterpret_cast<float4*>(calive + 4);                   // [256] original boxes
  uint64_t* ckey = reinterpret_cast<uint64_t*>(cobox + kChunk);            // [256]
  int32_t* clabel = reinterpret_cast<int32_t*>(ckey + kChunk);             // [256]
  float4* kbox = reinterpret_cast<float4*>(clabel + kChunk);               // [kKeptLds]
  float* karea = reinterpret_cast<float*>(kbox + kKeptLds);                // [kKeptLds]
  int32_t* kcount = reinterpret_cast<int32_t*>(karea + kKeptLds);          // [1]
  int16_t* csel = reinterpret_cast<int16_t*>(kcount + 4);                  // [256] kept candidates of the chunk, in order

  const float off_unit = p.max_coord[img] + 1.f;  // max_coordinate + 1 (mmcv batched_nms)
  const float* ibox = p.boxes + (size_t)img * p.boxes_per_image * 4;
  const int32_t* ilab = p.labels ? p.labels + (size_t)img * p.label_stride : nullptr;
  float* odet = p.out_dets + (size_t)img * p.max_out * 5;
  int32_t* olab = p.out_labels + (size_t)img * p.max_out;
  int64_t* oidx = p.out_index + (size_t)img * p.max_out;
  if (tid == 0) *kcount = 0;
  __syncthreads();

  // A chunk's candidates (key -> box: two dependent global reads) are fetched one chunk AHEAD into registers of the first
  // 256 threads: the ~3 us round trip used to open every chunk.
  float4 f_ob = make_float4(0.f, 0.f, 0.f, 0.f);
  uint64_t f_key = 0;
  int f_lab = 0;
  auto fetch = [&](int c0) {
    f_ob = make_float4(0.f, 0.f, 0.f, 0.f);
    f_key = 0;
    f_lab = 0;
    if (tid < kChunk && c0 + tid < n) {
      f_key = gkeys[c0 + tid];
      const uint32_t flat = (uint32_t)f_key;
      uint32_t bi;
      if (p.fused_classes > 0) {
        bi = flat / (uint32_t)p.fused_classes;
        f_lab = (int)(flat - bi * (uint32_t)p.fused_classes);
      } else {
        bi = flat;
        f_lab = ilab ? ilab[flat] : 0;
      }
      f_ob = reinterpret_cast<const float4*>(ibox)[bi];
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < n; c0 += kChunk) {
    const int cn = min(kChunk, n - c0);
    const int kept = *kcount;
    if (kept >= p.max_out || (YV4_ABLATE(p.ablate, 8) && c0 > 0)) break;
    // stage the chunk
    if (tid < kChunk) {
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      float ar = 0.f;
      if (tid < cn) {
        const float off = (float)f_lab * off_unit;  // idxs.to(boxes) * (max + 1)
        bb = make_float4(f_ob.x + off, f_ob.y + off, f_ob.z + off, f_ob.w + off);
        ar = (bb.z - bb.x) * (bb.w - bb.y);
      }
      cbox[tid] = bb;
      carea[tid] = ar;
      cobox[tid] = f_ob;
      ckey[tid] = f_key;
      clabel[tid] = f_lab;
    }
    __syncthreads();
    fetch(c0 + kChunk);
    // (a) chunk vs kept: thread -> (candidate i = tid & 255, quarter q = tid >> 8)
    {
      const int i = tid & (kChunk - 1);
      const int q = tid >> 8;
      bool dead = false;
      if (i < cn) {
        const float4 bj = cbox[i];
        const float aj = carea[i];
        for (int k = q; k < kept && !dead && !YV4_ABLATE(p.ablate, 2); k += 4) {
          float4 bk;
          float ak;
          if (k < kKeptLds) {
            bk = kbox[k];
            ak = karea[k];
          } else {
            // written earlier by this workgroup (one thread per kept box, after the walk of its chunk): read around the L1
            const volatile float* vd = odet;
            const volatile int32_t* vl = olab;
            const float off = (float)vl[k] * off_unit;
            bk = make_float4(vd[k * 5 + 0] + off, vd[k * 5 + 1] + off, vd[k * 5 + 2] + off, vd[k * 5 + 3] + off);
            ak = (bk.z - bk.x) * (bk.w - bk.y);
          }
          dead = iou_gt(bk, ak, bj, aj, p.iou_thr, p.iou_form);
        }
      } else {
        dead = true;
      }
      // combine the four quarters: every wave holds 64 candidates of one quarter
      const unsigned long long live = __ballot(!dead);
      if (q == 0 && (tid & 63) == 0) calive[tid >> 6] = live;
      __syncthreads();
      if (q != 0 && (tid & 63) == 0) atomicAnd(reinterpret_cast<unsigned long long*>(&calive[(tid & 255) >> 6]), live);
    }
    // (b) chunk x chunk bitmask: thread -> (row i = tid >> 2, word w = tid & 3)
    {
      const int i = tid >> 2;
      const int w = tid & 3;
      uint64_t bits = 0;
      if (i < cn) {
        const float4 bi = cbox[i];
        const float ai = carea[i];
        const int j0 = w * 64;
        for (int jj = 0; jj < 64 && !YV4_ABLATE(p.ablate, 4); ++jj) {
          const int j = j0 + jj;
          if (j > i && j < cn && iou_gt(bi, ai, cbox[j], carea[j], p.iou_thr, p.iou_form)) bits |= 1ull << jj;
        }
      }
      cmask[i * 4 + w] = bits;
    }
    __syncthreads();
    // (c) the greedy resolve is sequential, but only its DECISIONS are: the walk over the live bits records the kept
    // candidates of the chunk (a ctz, four mask words and a list entry per kept box); the outputs -- five floats,
    // label, index and the LDS copy per kept box -- are then written by one thread per kept box.  (With the stores inside
    // the walk a kept box cost ~30 dependent LDS / global operations: ~100 us of a 180 us launch at 300 kept per image.)
    if (tid < 64) {
      // Wave 0, all lanes in step: lane L holds the mask rows 4L .. 4L+3 in registers and the row of a kept candidate
      // comes by v_readlane (the candidate index is wave-uniform) instead of four dependent LDS reads per kept box.
      uint64_t rows[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int w = 0; w < 4; ++w) rows[r][w] = cmask[(4 * tid + r) * 4 + w];
      auto uni64 = [](uint64_t v, int lane) -> uint64_t {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
        return ((uint64_t)hi << 32) | lo;
      };
      uint64_t alive[4], removed[4] = {0, 0, 0, 0};
#pragma unroll
      for (int w = 0; w < 4; ++w) alive[w] = uni64(calive[w], 0);
      int k = kept;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        uint64_t cur = k < p.max_out ? alive[w] & ~removed[w] : 0ull;
        while (cur) {
          const int b = __builtin_ctzll(cur);
          const int i = w * 64 + b;
          if (tid == 0) csel[k - kept] = (int16_t)i;
          ++k;
          const int src = i >> 2;
          uint64_t m[4];
          switch (i & 3) {
            case 0:
#pragma unroll
              for (int ww = 0; ww < 4; ++ww) m[ww] = uni64(rows[0][ww], src);
              break;
            case 1:
#pragma unroll
              for (int ww = 0; ww < 4; ++ww) m[ww] = uni64(rows[1][ww], src);
              break;
            case 2:
#pragma unroll
              for (int ww = 0; ww < 4; ++ww) m[ww] = uni64(rows[2][ww], src);
              break;
            default:
#pragma unroll
              for (int ww = 0; ww < 4; ++ww) m[ww] = uni64(rows[3][ww], src);
              break;
          }
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) removed[ww] |= m[ww];
          const uint64_t above = b == 63 ? 0ull : (~0ull << (b + 1));
          cur = k < p.max_out ? alive[w] & ~removed[w] & above : 0ull;
        }
      }
      if (tid == 0) *kcount = k;
    }
    __syncthreads();
    {
      const int knew = *kcount;
      if (tid < knew - kept) {
        const int k = kept + tid;
        const int i = csel[tid];
        const float4 ob = cobox[i];
        const uint64_t key = ckey[i];
        odet[k * 5 + 0] = ob.x;
        odet[k * 5 + 1] = ob.y;
        odet[k * 5 + 2] = ob.z;
        odet[k * 5 + 3] = ob.w;
        odet[k * 5 + 4] = key_to_score((uint32_t)(key >> 32));
        olab[k] = clabel[i];
        oidx[k] = (int64_t)(uint32_t)key;
        if (k < kKeptLds) {
          kbox[k] = cbox[i];
          karea[k] = carea[i];
        }
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) p.out_count[img] = *kcount;
}

constexpr size_t kNmsLdsSort = (size_t)kSortCap * sizeof(uint64_t);
constexpr size_t kNmsLdsChunk = (size_t)kChunk * (16 + 4 + 32 + 16 + 8 + 4) + 32 + (size_t)kKeptLds * 20 + 16 + (size_t)kChunk * 2;
constexpr size_t kNmsLds = kNmsLdsSort > kNmsLdsChunk ? kNmsLdsSort : kNmsLdsChunk;

// keys / max for the standalone batched_nms op
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          int64_t n, uint64_t* keys, float* max_coord) {
  float m = -__builtin_huge_valf();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = ((uint64_t)score_to_key(scores[i]) << 32) | (uint32_t)i;
    const float4 b = reinterpret_cast<const float4*>(boxes)[i];
    m = fmaxf(m, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > -__builtin_huge_valf()) atomic_max_float(max_coord, m);
}

__global__ void nms_prepare_count_kernel(int32_t* counts, float* max_coord, int32_t n) {
  counts[0] = n;
  max_coord[0] = -__builtin_huge_valf();
}

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_decode_reset(int32_t* counts, float* max_coord, int N, void* stream) {
  YV4_REQUIRE(counts && max_coord && N > 0, "decode_reset: bad argument");
  hipLaunchKernelGGL(decode_reset_kernel, dim3((N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     counts, max_coord, N);
  YV4_CHECK_LAUNCH("decode_reset");
  return YV4_OK;
}

static int decode_impl(const yv4_level_desc* levels, int num_levels, int N, int A, int num_classes, float score_thr,
                       const float* scale_factor, float* boxes, float* conf, float* cls, uint64_t* keys,
                       int64_t key_cap, int32_t* counts, float* max_coord, const uint64_t* topk_keys, void* stream,
                       int v3, float conf_thr, int topk_per_level) {
  YV4_REQUIRE(levels && boxes && keys && counts && max_coord, "decode_filter: null pointer");
  YV4_REQUIRE(num_levels > 0 && num_levels <= kMaxLevels, "decode_filter: 1..%d levels supported", kMaxLevels);
  YV4_REQUIRE(N > 0 && N <= 65535 && A > 0 && A <= 8 && num_classes >= 0, "decode_filter: bad N/A/num_classes");
  YV4_REQUIRE(num_classes > 0 || cls == nullptr, "decode_filter: a class-agnostic head has no class scores to return");
  YV4_REQUIRE(key_cap > 0, "decode_filter: key_cap must be positive");
  YV4_REQUIRE(((uintptr_t)boxes & 15) == 0, "decode_filter: boxes must be 16-byte aligned");
  DecodeArgs a;
  long long total = 0;
  int blocks = 0;
  for (int l = 0; l < num_levels; ++l) {
    YV4_REQUIRE(levels[l].pred && levels[l].H > 0 && levels[l].W > 0 && levels[l].stride > 0,
                "decode_filter: level %d is malformed", l);
    a.pred[l] = levels[l].pred;
    a.H[l] = levels[l].H;
    a.W[l] = levels[l].W;
    a.stride[l] = levels[l].stride;
    a.level_base[l] = (int)total;
    a.block_base[l] = blocks;
    for (int k = 0; k < 8; ++k)
      for (int c = 0; c < 4; ++c) a.base[l][k][c] = levels[l].base_anchors[k][c];
    const long long nb = (long long)levels[l].H * levels[l].W * A;
    total += nb;
    blocks += (int)((nb + kDecBoxes - 1) / kDecBoxes);
  }
  a.block_base[num_levels] = blocks;
  YV4_REQUIRE(total * (num_classes > 0 ? num_classes : 1) < (1LL << 32),
              "decode_filter: anchors*classes = %lld overflows the 32-bit flat index", total * num_classes);
  a.num_levels = num_levels; a.N = N; a.A = A; a.C = num_classes; a.total_anchors = (int)total;
  a.score_thr = score_thr; a.scale_factor = scale_factor; a.boxes = boxes; a.conf = conf; a.cls = cls;
  a.keys = keys; a.key_cap = key_cap; a.counts = counts; a.max_coord = max_coord;
  a.topk = topk_keys;
  a.v3 = v3; a.conf_thr = conf_thr; a.topk_per_level = topk_per_level;
  static const int ablate = YV4_ENV_INT("YV4_DEC_ABLATE", 0);
  a.ablate = ablate;
  const size_t lds = (size_t)kDecBoxes * (5 + num_classes) * sizeof(float);
  // the kernel's static LDS (the key buffer, its counters, the per-wave maxima) shares the 64 KB a launch may use
  // without the opt-in attribute
  constexpr size_t kDecStaticLds = sizeof(uint64_t) * kDecKeyBuf + 3 * sizeof(int) + 4 * sizeof(float) + 64;
  YV4_REQUIRE(lds + kDecStaticLds <= 64 * 1024, "decode_filter: num_classes %d too large for the LDS tile (%zu + %zu bytes)",
              num_classes, lds, kDecStaticLds);
  hipLaunchKernelGGL(decode_filter_kernel, dim3((blocks + kDecTiles - 1) / kDecTiles, N), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), a);
  YV4_CHECK_LAUNCH("decode_filter");
  return YV4_OK;
}

extern "C" int yv4_decode_filter(const yv4_level_desc* levels, int num_levels, int N, int A, int num_classes,
                                 float score_thr, const float* scale_factor, float* boxes, float* conf, float* cls,
                                 uint64_t* keys, int64_t key_cap, int32_t* counts, float* max_coord,
                                 const uint64_t* topk_keys, void* stream) {
  return decode_impl(levels, num_levels, N, A, num_classes, score_thr, scale_factor, boxes, conf, cls, keys, key_cap,
                     counts, max_coord, topk_keys, stream, 0, 0.f, 0);
}

extern "C" int yv4_decode_filter_v3(const yv4_level_desc* levels, int num_levels, int N, int A, int num_classes,
                                    float score_thr, float conf_thr, const float* scale_factor, float* boxes,
                                    float* conf, float* cls, uint64_t* keys, int64_t key_cap, int32_t* counts,
                                    float* max_coord, const uint64_t* topk_keys_per_level, void* stream) {
  YV4_REQUIRE(num_classes > 0, "decode_filter_v3: num_classes must be positive");
  return decode_impl(levels, num_levels, N, A, num_classes, score_thr, scale_factor, boxes, conf, cls, keys, key_cap,
                     counts, max_coord, topk_keys_per_level, stream, 1, conf_thr, 1);
}

extern "C" int yv4_nms_images(uint64_t* keys, int64_t key_cap, const int32_t* counts, const float* max_coord,
                              const float* boxes, int64_t boxes_per_image, const int32_t* labels, int64_t label_stride,
                              int fused_classes, int N, float iou_thr, int max_out, int split_thr, float* out_dets,
                              int32_t* out_labels, int64_t* out_index, int32_t* out_count, void* stream) {
  YV4_REQUIRE(keys && counts && max_coord && boxes && out_dets && out_labels && out_index && out_count,
              "nms_images: null pointer");
  YV4_REQUIRE(N > 0 && max_out > 0 && key_cap > 0 && boxes_per_image > 0, "nms_images: bad sizes");
  YV4_REQUIRE(fused_classes >= 0, "nms_images: fused_classes must be >= 0");
  YV4_REQUIRE(((uintptr_t)boxes & 15) == 0, "nms_images: boxes must be 16-byte aligned");
  if (split_thr <= 0 || split_thr > kSortCap) split_thr = kSortCap + 1;
  NmsArgs a;
  a.keys = keys; a.key_cap = key_cap; a.counts = counts; a.max_coord = max_coord; a.boxes = boxes;
  a.boxes_per_image = boxes_per_image; a.labels = labels; a.label_stride = label_stride;
  a.fused_classes = fused_classes; a.iou_thr = iou_thr; a.iou_form = nms_iou_form(); a.max_out = max_out;
  static const int nms_ablate = YV4_ENV_INT("YV4_NMS_ABLATE", 0);
  a.ablate = nms_ablate; a.split_thr = split_thr;
  a.out_dets = out_dets; a.out_labels = out_labels; a.out_index = out_index; a.out_count = out_count;
  static LdsAttrOnce once;
  if (int rc = ensure_dyn_lds(once, reinterpret_cast<const void*>(nms_images_kernel), kNmsLds, "nms_images")) return rc;
  hipLaunchKernelGGL(nms_images_kernel, dim3(N), dim3(kNmsThreads), kNmsLds, reinterpret_cast<hipStream_t>(stream), a);
  YV4_CHECK_LAUNCH("nms_images");
  return YV4_OK;
}

extern "C" int yv4_nms_prepare(const float* boxes, const float* scores, int64_t n, uint64_t* keys, int32_t* counts,
                               float* max_coord, void* stream) {
  YV4_REQUIRE(keys && counts && max_coord, "nms_prepare: null pointer");
  YV4_REQUIRE(n >= 0 && n < (1LL << 31), "nms_prepare: n out of range");
  YV4_REQUIRE(n == 0 || (boxes && scores), "nms_prepare: null boxes/scores");
  YV4_REQUIRE(((uintptr_t)boxes & 15) == 0, "nms_prepare: boxes must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(nms_prepare_count_kernel, dim3(1), dim3(1), 0, s, counts, max_coord, (int32_t)n);
  if (n > 0) {
    unsigned g = (unsigned)((n + 255) / 256);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(nms_prepare_kernel, dim3(g), dim3(256), 0, s, boxes, scores, n, keys, max_coord);
  }
  YV4_CHECK_LAUNCH("nms_prepare");
  return YV4_OK;
}
