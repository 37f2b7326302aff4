// YOLOCSPHead training loss, forward and backward, in three kernels each way and without a host round trip:
//   mmdet/core/anchor/yolov4_anchor_generator.py:12-134   responsible_indices (neighbor = 2)
//   mmdet/models/dense_heads/yolocsp_head.py:384-575       loss / loss_single_no_assigner / get_targets_no_assigner
//   mmdet/core/bbox/coder/yolov4_bbox_coder.py:39-67       decode
//   mmdet/core/bbox/iou_calculators/iou2d_calculator.py    aligned GIoU (eps on union and on the enclosing area)
//   mmdet/models/losses/cross_entropy_loss.py:58-91        sigmoid BCE, mean
// The reference builds the positives with ~150 small tensor ops and two nonzero() host syncs per level, gathers
// them out of a dense fp32 (N, H*W*A, 5+C) copy of the prediction map and scatters the objectness targets with
// an index_put whose result for duplicate positives depends on the execution order.  Here:
//
//   assign   one thread per candidate slot (level, neighbour kind k, base anchor a, ground truth g): shape test,
//            neighbour-cell test, anchor index.  The slot number (k*A + a)*G + g is the position the positive has
//            in the reference's index lists (mask indexing is row-major), so "the last write wins" -- what the
//            reference's index_put does when it runs sequentially -- is an atomicMax of the slot number per
//            anchor box: deterministic, no compaction, no count needed on the host.
//   pos      one wavefront per valid slot: the 5+C logits of its anchor box straight from the head conv's raw
//            NHWC output (+ bias), decode, GIoU, class BCE; forward writes the objectness target of the slot and
//            adds to the loss sums (double); backward accumulates the row gradient into the winner slot's row.
//   dense    forward: objectness BCE over every anchor box, target = the winner slot's (or 0); backward: writes
//            the WHOLE gradient tensor of the conv output once (zeros, objectness gradients, the positive rows),
//            in the conv's own dtype and layout, and reduces the bias gradient on the way.
//
// Arithmetic is fp32 in the reference's expression order where a value is rounded (decode, GIoU); sums run in
// double.  Compiled with -ffp-contract=off.
#include "yv4_common.h"

namespace yv4 {

constexpr int kLossLevels = YV4_LOSS_MAX_LEVELS;

struct LossLv {
  const void* raw; void* draw; const float* bias; double* dbias;
  int H, W, Cp, stride;
  float base[8][4];
  long long anchor_off;   // first anchor box of the level inside an image
  long long block0;       // first workgroup of the level in the dense backward launch
  FastDiv fd_hw, fd_cpr;  // row / (H*W), chunk index / (Cp / chunk): the dense backward runs one divide per 16 bytes
};

struct LossArgs {
  LossLv lv[kLossLevels];
  int L, N, A, attr, C, G;
  long long TA;           // anchor boxes per image, all levels
  long long S;            // candidate slots per level = 5 * A * G
  FastDiv fd_TA, fd_A;
  const float* gt; const int64_t* gt_label; const int64_t* gt_img;
  float shape_thr, smooth, ratio, eps, w_cls, w_conf, w_bbox;
  int32_t* slot_anchor; int32_t* winner; int32_t* npos; float* conf_t; float* gpos; double* sums;
  const float* gout;
  float* losses;          // optional: (L, 3) float [cls | conf | bbox] written by the forward's last kernel
  int det;                // yv4_set_deterministic: sums / dbias / gpos are fixed-point words, [hi (n) | lo (n)] each
  long long gpos_n;       // L * S * attr
};

// loss sums (3 per level): doubles, or fixed-point words with the lo words 3*L entries further on
__device__ __forceinline__ void loss_sum_add(const LossArgs& p, int i, double v) {
  if (p.det) fx_add<kFxStat>(reinterpret_cast<u64_t*>(p.sums) + i, reinterpret_cast<u64_t*>(p.sums) + 3 * p.L + i, v);
  else atomicAdd(&p.sums[i], v);
}
// sums (doubles by now) + npos -> the (L, 3) losses: one thread per level
__global__ void loss_finish_kernel(LossArgs p) {
  const int l = threadIdx.x;
  if (l >= p.L) return;
  const double n = (double)p.npos[l];
  const double per_pos = n > 0.0 ? 1.0 / n : 0.0;
  const double boxes = (double)p.N * p.lv[l].H * p.lv[l].W * p.A;
  p.losses[l * 3 + 0] = (float)(p.sums[l * 3 + 0] * per_pos / (double)(p.C > 0 ? p.C : 1) * (double)p.w_cls);
  p.losses[l * 3 + 1] = (float)(p.sums[l * 3 + 1] / boxes * (double)p.w_conf);
  p.losses[l * 3 + 2] = (float)(p.sums[l * 3 + 2] * per_pos * (double)p.w_bbox);
}

template <int SHIFT>
__global__ void loss_fx_decode_kernel(double* __restrict__ buf, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64_t* w = reinterpret_cast<const u64_t*>(buf);
  buf[i] = fx_value<SHIFT>(w[i], w[n + i]);
}

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

__device__ __forceinline__ float rem1(float v) {      // torch's `v % 1.` (remainder: sign of the divisor)
  float r = fmodf(v, 1.f);
  if (r < 0.f) r += 1.f;
  return r;
}

__device__ __forceinline__ float bce_logits(float x, float t) {
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void yolo_assign_kernel(LossArgs p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.S * p.L) return;            // (tail lanes leave; __ballot() below only sees the lanes still here)
  const int ii = (int)i, S = (int)p.S;   // S * L < 2^31 (host)
  const int l = ii / S;
  const int s = ii - l * S;
  const int ag = s / p.G;
  const int g = s - ag * p.G;
  const int k = ag / p.A;
  const int a = ag - k * p.A;
  const LossLv& lv = p.lv[l];
  const float* b = p.gt + 4 * (size_t)g;
  const float cx = 0.5f * (b[2] + b[0]), cy = 0.5f * (b[3] + b[1]);
  const float gw = b[2] - b[0], gh = b[3] - b[1];
  const float bw = lv.base[a][2] - lv.base[a][0], bh = lv.base[a][3] - lv.base[a][1];
  float dw = gw / bw, dh = gh / bh;
  dw = fmaxf(dw, 1.f / dw);
  dh = fmaxf(dh, 1.f / dh);
  bool ok = fmaxf(dw, dh) < p.shape_thr;
  const float st = (float)lv.stride;
  const float x = cx / st, y = cy / st;
  const float ix = (float)lv.W - x, iy = (float)lv.H - y;
  float ox = 0.f, oy = 0.f;
  switch (k) {
    case 1: ok = ok && rem1(x) < 0.5f && x > 1.f; ox = -1.f; break;     // left
    case 2: ok = ok && rem1(y) < 0.5f && y > 1.f; oy = -1.f; break;     // up
    case 3: ok = ok && rem1(ix) < 0.5f && ix > 1.f; ox = 1.f; break;    // right
    case 4: ok = ok && rem1(iy) < 0.5f && iy > 1.f; oy = 1.f; break;    // down
    default: break;
  }
  const long long px = (long long)(x + ox), py = (long long)(y + oy);   // .long(): truncation
  const long long img = p.gt_img[g];
  ok = ok && px >= 0 && px < lv.W && py >= 0 && py < lv.H && img >= 0 && img < p.N;
  int anchor = -1;
  if (ok) {
    anchor = (int)((py * lv.W + px) * p.A + a);
    atomicMax(&p.winner[img * p.TA + lv.anchor_off + anchor], (int)s);
  }
  p.slot_anchor[i] = anchor;
  // positives per level: one atomic per wavefront (all of a level's positives hit the same counter)
  const int l0 = __builtin_amdgcn_readfirstlane(l);
  const unsigned long long same = __ballot(l == l0), live = __ballot(true);
  if (same == live) {
    const unsigned long long votes = __ballot(ok);
    if (votes && (int)(threadIdx.x & 63) == __ffsll((long long)live) - 1) atomicAdd(&p.npos[l0], __popcll(votes));
  } else if (ok) {
    atomicAdd(&p.npos[l], 1);
  }
}

// ---------------------------------------------------------------------------------------------------------
struct BoxTerms { float giou; float dt[4]; };

// decode + GIoU of one positive; with_grad: d(1 - giou)/d(box logits)
__device__ __forceinline__ BoxTerms box_terms(const float t[4], const LossLv& lv, int a, int gx, int gy, const float* tg,
                                              float eps, bool with_grad) {
  BoxTerms r;
  const float stride = (float)lv.stride;
  const float sx = (float)(gx * lv.stride), sy = (float)(gy * lv.stride);
  const float ax1 = lv.base[a][0] + sx, ay1 = lv.base[a][1] + sy, ax2 = lv.base[a][2] + sx, ay2 = lv.base[a][3] + sy;
  const float axc = (ax1 + ax2) * 0.5f, ayc = (ay1 + ay2) * 0.5f, aw = ax2 - ax1, ah = ay2 - ay1;
  float sg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) sg[j] = sigmoid_f32(t[j]);
  const float px = sg[0] * 2.f - 1.f, py = sg[1] * 2.f - 1.f;
  const float tw = sg[2] * 2.f, th = sg[3] * 2.f;
  const float xc = px * stride + axc, yc = py * stride + ayc;
  const float w = tw * tw * aw, h = th * th * ah;
  const float x1 = xc - w / 2.f, y1 = yc - h / 2.f, x2 = xc + w / 2.f, y2 = yc + h / 2.f;
  const float tx1 = tg[0], ty1 = tg[1], tx2 = tg[2], ty2 = tg[3];
  const float w1 = x2 - x1, h1 = y2 - y1;
  const float area1 = w1 * h1, area2 = (tx2 - tx1) * (ty2 - ty1);
  const float dwr = fminf(x2, tx2) - fmaxf(x1, tx1), dhr = fminf(y2, ty2) - fmaxf(y1, ty1);
  const float iw = fmaxf(dwr, 0.f), ih = fmaxf(dhr, 0.f);
  const float ov = iw * ih;
  const float Uraw = area1 + area2 - ov;
  const float U = fmaxf(Uraw, eps);
  const float ewr = fmaxf(x2, tx2) - fminf(x1, tx1), ehr = fmaxf(y2, ty2) - fminf(y1, ty1);
  const float ew = fmaxf(ewr, 0.f), eh = fmaxf(ehr, 0.f);
  const float Eraw = ew * eh;
  const float E = fmaxf(Eraw, eps);
  r.giou = ov / U - (E - U) / E;
  if (!with_grad) return r;
  // L = 1 - giou
  const float d_iou = -1.f, d_frac = 1.f;
  float d_ov = d_iou / U;
  float d_U = -d_iou * ov / (U * U) - d_frac / E;
  const float d_E = d_frac * U / (E * E);
  const float d_Uraw = d_U * (Uraw > eps ? 1.f : (Uraw == eps ? 0.5f : 0.f));
  d_ov -= d_Uraw;
  const float d_area1 = d_Uraw;
  const float d_Eraw = d_E * (Eraw > eps ? 1.f : (Eraw == eps ? 0.5f : 0.f));
  const float d_ewr = ewr >= 0.f ? d_Eraw * eh : 0.f, d_ehr = ehr >= 0.f ? d_Eraw * ew : 0.f;
  const float d_dwr = dwr >= 0.f ? d_ov * ih : 0.f, d_dhr = dhr >= 0.f ? d_ov * iw : 0.f;
  // max / min against the (constant) target: the gradient goes to the prediction where it is selected, half on a tie
  auto sel_gt = [](float a_, float b_) { return a_ > b_ ? 1.f : (a_ == b_ ? 0.5f : 0.f); };
  auto sel_lt = [](float a_, float b_) { return a_ < b_ ? 1.f : (a_ == b_ ? 0.5f : 0.f); };
  float d_x1 = -d_area1 * h1 - d_ewr * sel_lt(x1, tx1) - d_dwr * sel_gt(x1, tx1);
  float d_x2 = d_area1 * h1 + d_ewr * sel_gt(x2, tx2) + d_dwr * sel_lt(x2, tx2);
  float d_y1 = -d_area1 * w1 - d_ehr * sel_lt(y1, ty1) - d_dhr * sel_gt(y1, ty1);
  float d_y2 = d_area1 * w1 + d_ehr * sel_gt(y2, ty2) + d_dhr * sel_lt(y2, ty2);
  const float d_xc = d_x1 + d_x2, d_yc = d_y1 + d_y2;
  const float d_w = (d_x2 - d_x1) * 0.5f, d_h = (d_y2 - d_y1) * 0.5f;
  const float d_s0 = d_xc * stride * 2.f, d_s1 = d_yc * stride * 2.f;
  const float d_s2 = d_w * aw * 2.f * tw * 2.f, d_s3 = d_h * ah * 2.f * th * 2.f;
  r.dt[0] = d_s0 * sg[0] * (1.f - sg[0]);
  r.dt[1] = d_s1 * sg[1] * (1.f - sg[1]);
  r.dt[2] = d_s2 * sg[2] * (1.f - sg[2]);
  r.dt[3] = d_s3 * sg[3] * (1.f - sg[3]);
  return r;
}

constexpr int kSlotsPerWave = 8;

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void yolo_pos_kernel(LossArgs p) {
  const int lane = threadIdx.x & 63;
  const int S = (int)p.S, total = S * p.L;
  const int ws0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kSlotsPerWave;
  float part_cls = 0.f, part_box = 0.f;     // forward: this wave's sums for level part_l (lane 0)
  int part_l = -1;
  for (int ws = ws0; ws < ws0 + kSlotsPerWave && ws < total; ++ws) {
  const int anchor = p.slot_anchor[ws];
  if (anchor < 0) continue;                                 // wave-uniform
  const int l = ws / S;
  const int s = ws - l * S;
  const int g = s % p.G;
  const LossLv& lv = p.lv[l];
  const int a = anchor % p.A;
  const int cell = anchor / p.A;
  const int gx = cell % lv.W, gy = cell / lv.W;
  const long long img = p.gt_img[g];
  const T* row = reinterpret_cast<const T*>(lv.raw) + ((size_t)(img * lv.H + gy) * lv.W + gx) * lv.Cp + (size_t)a * p.attr;
  const float* bias = lv.bias + a * p.attr;
  float t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = ldf(row + j) + bias[j];
  const BoxTerms bt = box_terms(t, lv, a, gx, gy, p.gt + 4 * (size_t)g, p.eps, BWD);
  const int label = p.C > 0 ? (int)p.gt_label[g] : -1;
  const float t_on = p.smooth != 0.f ? (1.f - p.smooth) + p.smooth / (float)p.C : 1.f;
  const float t_off = p.smooth != 0.f ? p.smooth / (float)p.C : 0.f;
  if (!BWD) {
    float acc = 0.f;
    for (int c = lane; c < p.C; c += 64) acc += bce_logits(ldf(row + 5 + c) + bias[5 + c], c == label ? t_on : t_off);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
      if (l != part_l) {          // a wave's slots rarely straddle two levels: flush, then start the new level
        if (part_l >= 0) {
          if (p.C > 0) loss_sum_add(p, part_l * 3 + 0, (double)part_cls);
          loss_sum_add(p, part_l * 3 + 2, (double)part_box);
        }
        part_l = l; part_cls = 0.f; part_box = 0.f;
      }
      part_cls += acc;
      const float gl = 1.f - bt.giou;                        // the GIoU loss of the positive
      part_box += gl;
      const float q = fminf(fmaxf(1.f - gl, 0.f), 1.f);      // (1 - giou_loss).clamp(0, 1)
      p.conf_t[ws] = (1.f - p.ratio) + p.ratio * q;
    }
  } else {
    const int np = p.npos[l];
    const float k_box = p.gout[l * 3 + 2] * p.w_bbox / (float)np;
    const int wslot = p.winner[img * p.TA + lv.anchor_off + anchor];
    const size_t grow0 = ((size_t)l * S + wslot) * p.attr;
    float* grow = p.gpos + grow0;
    // several positives of one anchor box add into the winner's row: float atomics in arrival order, or -- deterministic
    // mode -- integer atomics on fixed-point words (gpos is then 2 * gpos_n 64-bit words)
    u64_t* ghi = reinterpret_cast<u64_t*>(p.gpos) + grow0;
    u64_t* glo = ghi + p.gpos_n;
    if (lane < 4) {
      if (p.det) fx_add<kFxGrad>(ghi + lane, glo + lane, (double)(bt.dt[lane] * k_box));
      else atomicAdd(&grow[lane], bt.dt[lane] * k_box);
    }
    if (p.C > 0) {
      const float k_cls = p.gout[l * 3 + 0] * p.w_cls / ((float)np * (float)p.C);
      for (int c = lane; c < p.C; c += 64) {
        const float x = ldf(row + 5 + c) + bias[5 + c];
        const float gv = (sigmoid_f32(x) - (c == label ? t_on : t_off)) * k_cls;
        if (p.det) fx_add<kFxGrad>(ghi + 5 + c, glo + 5 + c, (double)gv);
        else atomicAdd(&grow[5 + c], gv);
      }
    }
  }
  }   // slots of this wave
  if (!BWD && lane == 0 && part_l >= 0) {
    if (p.C > 0) loss_sum_add(p, part_l * 3 + 0, (double)part_cls);
    loss_sum_add(p, part_l * 3 + 2, (double)part_box);
  }
}

// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void yolo_dense_fwd_kernel(LossArgs p) {
  __shared__ double part[2 * kLossLevels];     // deterministic mode: hi words, then lo words
  if (threadIdx.x < 2 * kLossLevels) part[threadIdx.x] = 0.0;
  __syncthreads();
  u64_t* pw = reinterpret_cast<u64_t*>(part);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < p.TA * p.N) {
    const int n = fd_div((int)i, p.fd_TA);                 // N * TA < 2^31 (host)
    const int j = (int)i - n * (int)p.TA;
    int l = 0;
    while (l + 1 < p.L && j >= (int)p.lv[l + 1].anchor_off) ++l;
    const LossLv& lv = p.lv[l];
    const int jl = j - (int)lv.anchor_off;
    const int cell = fd_div(jl, p.fd_A);
    const int a = jl - cell * p.A;
    const int c = a * p.attr + 4;
    const float x = ldf(reinterpret_cast<const T*>(lv.raw) + ((size_t)n * lv.H * lv.W + cell) * lv.Cp + c) + lv.bias[c];
    const int w = p.winner[i];
    const float tgt = w >= 0 ? p.conf_t[(size_t)l * p.S + w] : 0.f;
    if (p.det) fx_add<kFxStat>(pw + l, pw + kLossLevels + l, (double)bce_logits(x, tgt));
    else atomicAdd(&part[l], (double)bce_logits(x, tgt));
  }
  __syncthreads();
  if (p.det) {
    if (threadIdx.x < p.L) {
      u64_t* g = reinterpret_cast<u64_t*>(p.sums);
      const u64_t h = pw[threadIdx.x], lo = pw[kLossLevels + threadIdx.x];
      const int i = threadIdx.x * 3 + 1;
      if (h) atomicAdd(g + i, h);
      if (lo >> 63) atomicOr(g + 3 * p.L + i, 1ull << 63);
      if (lo & ~(1ull << 63)) atomicAdd(g + 3 * p.L + i, lo & ~(1ull << 63));
    }
    return;
  }
  if (threadIdx.x < p.L && part[threadIdx.x] != 0.0) atomicAdd(&p.sums[threadIdx.x * 3 + 1], part[threadIdx.x]);
}

template <typename T> struct Chunk;
template <> struct Chunk<float> { static constexpr int n = 4; };
template <> struct Chunk<_Float16> { static constexpr int n = 8; };
template <> struct Chunk<__bf16> { static constexpr int n = 8; };

template <typename T>
__global__ __launch_bounds__(256) void yolo_dense_bwd_kernel(LossArgs p) {
  constexpr int CH = Chunk<T>::n;
  extern __shared__ float db[];                      // [Cp] bias-gradient partials of the workgroup (det: [2][Cp] words)
  u64_t* dbw = reinterpret_cast<u64_t*>(db);
  int l = 0;
  while (l + 1 < p.L && (long long)blockIdx.x >= p.lv[l + 1].block0) ++l;
  const LossLv& lv = p.lv[l];
  if (p.det) for (int c = threadIdx.x; c < 2 * lv.Cp; c += 256) dbw[c] = 0;
  else for (int c = threadIdx.x; c < lv.Cp; c += 256) db[c] = 0.f;
  __syncthreads();
  const int cpr = lv.Cp / CH;
  const int rows = p.N * lv.H * lv.W;                      // rows * cpr < 2^31 (host)
  // a level's workgroups stride over its 16-byte chunks: few, long-lived workgroups keep the number of
  // same-address bias-gradient atomics at the end small (one per workgroup per channel: with one workgroup per
  // 256 chunks the three objectness channels saw 46 k serialised atomics each, 650 us of a 60 us pass)
  const int nblk = (int)((l + 1 < p.L ? p.lv[l + 1].block0 : (long long)gridDim.x) - lv.block0);
  for (int idx = (int)(((long long)blockIdx.x - lv.block0) * 256 + threadIdx.x); idx < rows * cpr; idx += nblk * 256) {
  const int row = fd_div(idx, lv.fd_cpr);
  {
    const int c0 = (idx - row * cpr) * CH;
    const int HW = lv.H * lv.W;
    const int n = fd_div(row, lv.fd_hw), cell = row - n * HW;
    const float k_conf = p.gout[l * 3 + 1] * p.w_conf / (float)((long long)p.N * HW * p.A);
    const T* src = reinterpret_cast<const T*>(lv.raw) + (size_t)row * lv.Cp;
    const int32_t* win = p.winner + (long long)n * p.TA + lv.anchor_off + (long long)cell * p.A;
    float v[CH];
    int a_cached = -1, w_cached = -1;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = c0 + u;
      v[u] = 0.f;
      if (c >= p.A * p.attr) continue;
      const int a = c / p.attr, j = c - a * p.attr;
      if (a != a_cached) { a_cached = a; w_cached = win[a]; }
      if (j == 4) {
        const float x = ldf(src + c) + lv.bias[c];
        const float tgt = w_cached >= 0 ? p.conf_t[(size_t)l * p.S + w_cached] : 0.f;
        v[u] = (sigmoid_f32(x) - tgt) * k_conf;
      } else if (w_cached >= 0) {
        const size_t gi = ((size_t)l * p.S + w_cached) * p.attr + j;
        if (p.det) {
          const u64_t* gw = reinterpret_cast<const u64_t*>(p.gpos);
          v[u] = (float)fx_value<kFxGrad>(gw[gi], gw[p.gpos_n + gi]);
        } else {
          v[u] = p.gpos[gi];
        }
      }
      if (v[u] != 0.f) {
        if (p.det) fx_add<kFxGrad>(dbw + c, dbw + lv.Cp + c, (double)v[u]);
        else atomicAdd(&db[c], v[u]);
      }
    }
    T* dst = reinterpret_cast<T*>(lv.draw) + (size_t)row * lv.Cp + c0;
    alignas(16) T o[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) o[u] = (T)v[u];
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
  }
  }   // chunks of this workgroup
  __syncthreads();
  if (p.det) {        // dbias: [hi (A*attr) | lo (A*attr)] words, decoded in place after the launch
    u64_t* g = reinterpret_cast<u64_t*>(lv.dbias);
    const int nb = p.A * p.attr;
    for (int c = threadIdx.x; c < nb; c += 256) {
      const u64_t h = dbw[c], lo = dbw[lv.Cp + c];
      if (h) atomicAdd(g + c, h);
      if (lo >> 63) atomicOr(g + nb + c, 1ull << 63);
      if (lo & ~(1ull << 63)) atomicAdd(g + nb + c, lo & ~(1ull << 63));
    }
    return;
  }
  for (int c = threadIdx.x; c < p.A * p.attr; c += 256)
    if (db[c] != 0.f) atomicAdd(&lv.dbias[c], (double)db[c]);
}

static int fill_args(const yv4_loss_desc* d, LossArgs& a, const char* who) {
  YV4_REQUIRE(d, "%s: null descriptor", who);
  YV4_REQUIRE(d->num_levels >= 1 && d->num_levels <= kLossLevels, "%s: 1..%d levels", who, kLossLevels);
  YV4_REQUIRE(d->N > 0 && d->A >= 1 && d->A <= 8 && d->num_classes >= 0 && d->G >= 0, "%s: bad sizes", who);
  YV4_REQUIRE(d->dtype == YV4_F32 || d->dtype == YV4_F16 || d->dtype == YV4_BF16, "%s: dtype must be f32, f16 or bf16", who);
  YV4_REQUIRE(d->winner && d->npos && d->sums, "%s: work buffers missing", who);
  YV4_REQUIRE(d->G == 0 || (d->gt && d->gt_img && d->slot_anchor && d->conf_t && (d->num_classes == 0 || d->gt_label)),
              "%s: ground-truth tables missing", who);
  a = LossArgs{};
  a.L = d->num_levels; a.N = d->N; a.A = d->A; a.C = d->num_classes; a.attr = 5 + d->num_classes; a.G = d->G;
  a.S = 5LL * d->A * d->G;
  const int ch = d->dtype == YV4_F32 ? 4 : 8;
  long long off = 0;
  for (int l = 0; l < a.L; ++l) {
    const yv4_loss_level& s = d->levels[l];
    YV4_REQUIRE(s.raw && s.bias && s.H > 0 && s.W > 0 && s.stride > 0, "%s: level %d incomplete", who, l);
    YV4_REQUIRE(s.Cp % ch == 0 && s.Cp >= a.A * a.attr, "%s: level %d: pixel stride %d must be a multiple of %d and hold %d channels",
                who, l, s.Cp, ch, a.A * a.attr);
    YV4_REQUIRE(((uintptr_t)s.raw & 15) == 0, "%s: level %d: map must be 16-byte aligned", who, l);
    LossLv& t = a.lv[l];
    t.raw = s.raw; t.draw = s.draw; t.bias = s.bias; t.dbias = s.dbias;
    t.H = s.H; t.W = s.W; t.Cp = s.Cp; t.stride = s.stride;
    for (int k = 0; k < 8; ++k)
      for (int c = 0; c < 4; ++c) t.base[k][c] = s.base_anchors[k][c];
    t.anchor_off = off;
    off += (long long)s.H * s.W * a.A;
    t.fd_hw = make_fastdiv((unsigned)(s.H * s.W));
    t.fd_cpr = make_fastdiv((unsigned)(s.Cp / ch));
    // the dense backward walks 16-byte chunks with a 32-bit index that advances by up to 2048 * 256 per iteration
    YV4_REQUIRE((long long)d->N * s.H * s.W * (s.Cp / ch) < (1LL << 31) - 2048LL * 256, "%s: level %d: map too large", who, l);
  }
  a.TA = off;
  a.fd_TA = make_fastdiv((unsigned)off);
  a.fd_A = make_fastdiv((unsigned)a.A);
  YV4_REQUIRE(a.TA * a.N < (1LL << 31) && a.S * a.L < (1LL << 31), "%s: index space exceeds 31 bits", who);
  a.gt = d->gt; a.gt_label = d->gt_label; a.gt_img = d->gt_img;
  a.shape_thr = d->shape_thr; a.smooth = d->smooth; a.ratio = d->ratio; a.eps = d->eps;
  a.w_cls = d->w_cls; a.w_conf = d->w_conf; a.w_bbox = d->w_bbox;
  a.slot_anchor = d->slot_anchor; a.winner = d->winner; a.npos = d->npos; a.conf_t = d->conf_t; a.gpos = d->gpos;
  a.sums = d->sums;
  a.losses = d->losses;
  a.det = deterministic() ? 1 : 0;
  a.gpos_n = a.S * a.L * a.attr;
  return YV4_OK;
}

#define YV4_LOSS_DISPATCH(dtype, ...)                                   \
  do {                                                                  \
    if ((dtype) == YV4_F32) { using T = float; __VA_ARGS__; }           \
    else if ((dtype) == YV4_F16) { using T = _Float16; __VA_ARGS__; }   \
    else { using T = __bf16; __VA_ARGS__; }                             \
  } while (0)

}  // namespace yv4

using namespace yv4;

extern "C" int yv4_yolo_loss_fwd(const yv4_loss_desc* d, void* stream) {
  LossArgs a;
  if (int rc = fill_args(d, a, "yolo_loss_fwd")) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  bool ok = hipMemsetAsync(a.winner, 0xFF, sizeof(int32_t) * a.TA * a.N, s) == hipSuccess;
  ok = ok && hipMemsetAsync(a.npos, 0, sizeof(int32_t) * a.L, s) == hipSuccess;
  ok = ok && hipMemsetAsync(a.sums, 0, sizeof(double) * (a.det ? 6 : 3) * a.L, s) == hipSuccess;
  if (!ok) { set_error("yolo_loss_fwd: memset failed"); return YV4_E_LAUNCH; }
  const long long slots = a.S * a.L;
  if (slots > 0) {
    hipLaunchKernelGGL(yolo_assign_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, a);
    YV4_LOSS_DISPATCH(d->dtype, hipLaunchKernelGGL((yolo_pos_kernel<T, false>), dim3((unsigned)((slots + 4 * kSlotsPerWave - 1) / (4 * kSlotsPerWave))),
                                                  dim3(256), 0, s, a));
  }
  const long long boxes = a.TA * a.N;
  YV4_LOSS_DISPATCH(d->dtype, hipLaunchKernelGGL(yolo_dense_fwd_kernel<T>, dim3((unsigned)((boxes + 255) / 256)), dim3(256), 0,
                                                 s, a));
  if (a.det) hipLaunchKernelGGL(loss_fx_decode_kernel<kFxStat>, dim3(1), dim3(256), 0, s, a.sums, 3 * a.L);
  if (a.losses) hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, a);
  YV4_CHECK_LAUNCH("yolo_loss_fwd");
  return YV4_OK;
}

extern "C" int yv4_yolo_loss_bwd(const yv4_loss_desc* d, const float* grad_out, void* stream) {
  LossArgs a;
  if (int rc = fill_args(d, a, "yolo_loss_bwd")) return rc;
  YV4_REQUIRE(grad_out, "yolo_loss_bwd: grad_out missing");
  YV4_REQUIRE(a.S == 0 || a.gpos, "yolo_loss_bwd: gpos missing");
  a.gout = grad_out;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int ch = d->dtype == YV4_F32 ? 4 : 8;
  long long blocks = 0;
  int max_cp = 0;
  bool ok = true;
  for (int l = 0; l < a.L; ++l) {
    YV4_REQUIRE(a.lv[l].draw && a.lv[l].dbias, "yolo_loss_bwd: level %d: draw / dbias missing", l);
    YV4_REQUIRE(((uintptr_t)a.lv[l].draw & 15) == 0, "yolo_loss_bwd: level %d: draw must be 16-byte aligned", l);
    a.lv[l].block0 = blocks;
    const long long chunks = (long long)a.N * a.lv[l].H * a.lv[l].W * (a.lv[l].Cp / ch);
    long long nb = (chunks + 255) / 256;
    if (nb > 2048) nb = 2048;
    blocks += nb;
    if (a.lv[l].Cp > max_cp) max_cp = a.lv[l].Cp;
    ok = ok && hipMemsetAsync(a.lv[l].dbias, 0, sizeof(double) * (a.det ? 2 : 1) * a.A * a.attr, s) == hipSuccess;
  }
  const long long slots = a.S * a.L;
  if (slots > 0) ok = ok && hipMemsetAsync(a.gpos, 0, (a.det ? 16 : sizeof(float)) * slots * a.attr, s) == hipSuccess;
  if (!ok) { set_error("yolo_loss_bwd: memset failed"); return YV4_E_LAUNCH; }
  YV4_REQUIRE(blocks < (1LL << 31), "yolo_loss_bwd: too many workgroups");
  if (slots > 0)
    YV4_LOSS_DISPATCH(d->dtype, hipLaunchKernelGGL((yolo_pos_kernel<T, true>), dim3((unsigned)((slots + 4 * kSlotsPerWave - 1) / (4 * kSlotsPerWave))),
                                                  dim3(256), 0, s, a));
  YV4_LOSS_DISPATCH(d->dtype, hipLaunchKernelGGL(yolo_dense_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256),
                                                 (a.det ? 16 : sizeof(float)) * max_cp, s, a));
  if (a.det)
    for (int l = 0; l < a.L; ++l)
      hipLaunchKernelGGL(loss_fx_decode_kernel<kFxGrad>, dim3((a.A * a.attr + 255) / 256), dim3(256), 0, s, a.lv[l].dbias,
                         a.A * a.attr);
  YV4_CHECK_LAUNCH("yolo_loss_bwd");
  return YV4_OK;
}
