// Optimizer-side kernels of the training step on gfx950, all HBM-bound streaming passes over
// flat fp32 arenas (every parameter / gradient / momentum buffer / EMA copy of the model is a
// slice of one allocation, see flat_state.py).
//
// What they replace in the reference's step:
//   torch.optim.SGD(nesterov) with ONE PARAM GROUP PER PARAMETER (the warm-up hook demands it,
//     core/custom_hooks/warmup_hooks.py:24-32): ~430 groups x ~6 tiny kernels per step
//   GradScaler.unscale_ + clip_grad_norm_(35, L2)       (accum_optim_hooks.py:45-58)
//   StateEMAHook.after_train_iter: a Python loop of mul_/add_ over 658 state entries
//     (core/custom_hooks/ema_hooks.py:80-98)
// Here: one reduction pass (sum of squares + non-finite flag), one update pass with a per-segment
// hyper-parameter table, one EMA pass.  No host synchronisation anywhere: the clip coefficient,
// the skip-on-overflow decision and the loss-scale update stay on the device.
#include "yv4_common.h"

namespace yv4 {

// ---------------------------------------------------------------------------------
// sum of squares (double) + count of non-finite values of a flat fp32 array
// ---------------------------------------------------------------------------------
constexpr int kSumsqMaxWg = YV4_GRAD_PREPARE_MAX_WG;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n4, double* __restrict__ work,
                                                    int det) {
  // four independent 16-byte loads in flight per thread and four accumulators (one load per trip was a chain of memory
  // round trips: 214 us for the 256 MB of YOLOv4-L's gradients), one atomic per WORKGROUP (device-scope double atomics on
  // one word execute at the memory side, one after the other)
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  unsigned bad = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + u * stride;
      v[u] = j < n4 ? g4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // non-finite is decided on the VALUES (what GradScaler's unscale_ inspects): a finite gradient whose fp32
      // square would overflow must not skip the step; the squares are formed and summed in double
      bad += !(fabsf(v[u].x) <= 3.402823466e38f && fabsf(v[u].y) <= 3.402823466e38f && fabsf(v[u].z) <= 3.402823466e38f &&
               fabsf(v[u].w) <= 3.402823466e38f);
      acc[u] += (double)v[u].x * (double)v[u].x + (double)v[u].y * (double)v[u].y + (double)v[u].z * (double)v[u].z +
                (double)v[u].w * (double)v[u].w;
    }
  }
  double a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_down(a, o);
    bad += __shfl_down(bad, o);
  }
  __shared__ double wsum[4];
  __shared__ unsigned wbad[4];
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = a; wbad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    // deterministic mode: the workgroup's partial goes to its own slot, grad_ctrl_kernel adds the slots in index order
    if (det) work[2 + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    else atomicAdd(&work[0], (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
    const unsigned b = wbad[0] + wbad[1] + wbad[2] + wbad[3];
    if (b) atomicAdd(&work[1], (double)b);
  }
}

// ctrl[0] = multiplier applied to every gradient in the update (1/scale * clip coefficient)
// ctrl[1] = total L2 norm of the unscaled gradients (what the reference logs as grad_norm)
// ctrl[2] = 1 if any gradient is non-finite (the update and the EMA of this step are skipped)
// ctrl[3] = 1/scale used
__global__ __launch_bounds__(64) void grad_ctrl_kernel(const double* __restrict__ work, const float* __restrict__ scale_state,
                                                       float max_norm, float* __restrict__ ctrl, int det_slots) {
  // one wave.  det_slots > 0: the sum of squares is the per-workgroup partials work[2 ..], each lane adding its slots in
  // index order and the lanes combined by a fixed butterfly -- the same bits every run
  double ss = work[0];
  if (det_slots > 0) {
    ss = 0.0;
    for (int i = threadIdx.x; i < det_slots; i += 64) ss += work[2 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  }
  if (threadIdx.x != 0) return;
  const float inv_scale = scale_state ? 1.f / scale_state[0] : 1.f;
  const bool bad = work[1] > 0.0;
  const float norm = (float)sqrt(ss) * inv_scale;
  float coef = 1.f;
  if (max_norm > 0.f) {
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    coef = max_norm / (norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  ctrl[0] = inv_scale * coef;
  ctrl[1] = norm;
  ctrl[2] = bad ? 1.f : 0.f;
  ctrl[3] = inv_scale;
}

// torch.cuda.amp.GradScaler.update (dynamic loss scale): state = {scale, growth_tracker}
__global__ void scale_update_kernel(float* __restrict__ state, const float* __restrict__ ctrl, float growth, float backoff,
                                    int interval) {
  if (ctrl[2] != 0.f) {
    state[0] *= backoff;
    state[1] = 0.f;
  } else {
    const float t = state[1] + 1.f;
    if (t >= (float)interval) {
      state[0] *= growth;
      state[1] = 0.f;
    } else {
      state[1] = t;
    }
  }
}

// ---------------------------------------------------------------------------------
// SGD with momentum / Nesterov / weight decay, per-segment hyper-parameters.
//   g  = grad * ctrl[0] + wd * p
//   b  = mom * b + g                (dampening 0; a zero-initialised buffer reproduces torch's
//                                    "first step: b = g")
//   p -= lr * (nesterov ? g + mom * b : b)
// seg_off[s] .. seg_off[s+1] (floats, multiples of 4) is segment s; hyper[s] = {lr, momentum,
// weight_decay, nesterov}.  Each thread finds its segment by bisection over offsets in LDS.
// ---------------------------------------------------------------------------------
constexpr int kMaxSeg = 4096;

__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ b, int64_t n4,
                                                       const int64_t* __restrict__ seg_off,
                                                       const float4* __restrict__ hyper, int nseg,
                                                       const float* __restrict__ ctrl) {
  __shared__ int64_t off[kMaxSeg + 1];
  if (ctrl && ctrl[2] != 0.f) return;   // non-finite gradients: GradScaler.step skips the update
  for (int i = threadIdx.x; i <= nseg; i += 256) off[i] = seg_off[i];
  __syncthreads();
  const float mul = ctrl ? ctrl[0] : 1.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t e = i * 4;
    int lo = 0, hi = nseg;   // invariant: off[lo] <= e < off[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (off[mid] <= e) lo = mid; else hi = mid;
    }
    const float4 h = hyper[lo];
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = reinterpret_cast<float4*>(b)[i];
    float* pp = reinterpret_cast<float*>(&pv);
    const float* gp = reinterpret_cast<const float*>(&gv);
    float* bp = reinterpret_cast<float*>(&bv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gr = gp[k] * mul;
      if (h.z != 0.f) gr = gr + h.z * pp[k];
      const float nb = h.y * bp[k] + gr;
      bp[k] = nb;
      const float step = h.w != 0.f ? gr + h.y * nb : nb;
      pp[k] = pp[k] - h.x * step;
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(b)[i] = bv;
  }
}

// ema = m * ema + (1 - m) * x     (ema_hooks.py:93-95)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ x, int64_t n4, float m,
                                                  float one_minus_m) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 e = reinterpret_cast<float4*>(ema)[i];
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    e.x = e.x * m + v.x * one_minus_m;
    e.y = e.y * m + v.y * one_minus_m;
    e.z = e.z * m + v.z * one_minus_m;
    e.w = e.w * m + v.w * one_minus_m;
    reinterpret_cast<float4*>(ema)[i] = e;
  }
}

static inline unsigned stream_grid(int64_t n4) {
  const int64_t blocks = (n4 + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks));
}

}  // namespace yv4

using namespace yv4;

extern "C" {

int yv4_grad_prepare(const float* grad, int64_t n, const float* scale_state, float max_norm, double* work, float* ctrl,
                     void* stream) {
  YV4_REQUIRE(grad && work && ctrl, "grad_prepare: null pointer");
  YV4_REQUIRE(n >= 0 && n % 4 == 0, "grad_prepare: n must be a non-negative multiple of 4 (got %lld)", (long long)n);
  YV4_REQUIRE(((uintptr_t)grad & 15) == 0, "grad_prepare: grad must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(work, 0, 2 * sizeof(double), s) != hipSuccess) {
    set_error("grad_prepare: memset failed");
    return YV4_E_LAUNCH;
  }
  const int det = deterministic() ? 1 : 0;
  unsigned grid = 0;
  if (n > 0) {
    grid = stream_grid((n / 4 + 3) / 4);          // four 16-byte words per thread and trip
    if (grid > (unsigned)kSumsqMaxWg) grid = kSumsqMaxWg;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, grad, n / 4, work, det);
  }
  hipLaunchKernelGGL(grad_ctrl_kernel, dim3(1), dim3(64), 0, s, work, scale_state, max_norm, ctrl, det ? (int)grid : 0);
  YV4_CHECK_LAUNCH("grad_prepare");
  return YV4_OK;
}

int yv4_loss_scale_update(float* scale_state, const float* ctrl, float growth_factor, float backoff_factor,
                          int growth_interval, void* stream) {
  YV4_REQUIRE(scale_state && ctrl, "loss_scale_update: null pointer");
  YV4_REQUIRE(growth_factor > 1.f && backoff_factor > 0.f && backoff_factor < 1.f && growth_interval > 0,
              "loss_scale_update: need growth > 1, 0 < backoff < 1, interval > 0");
  hipLaunchKernelGGL(scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale_state, ctrl, growth_factor,
                     backoff_factor, growth_interval);
  YV4_CHECK_LAUNCH("loss_scale_update");
  return YV4_OK;
}

int yv4_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, const int64_t* seg_off,
                 const float* seg_hyper, int nseg, const float* ctrl, void* stream) {
  YV4_REQUIRE(param && grad && momentum_buf && seg_off && seg_hyper, "sgd_step: null pointer");
  YV4_REQUIRE(n >= 0 && n % 4 == 0, "sgd_step: n must be a non-negative multiple of 4 (got %lld)", (long long)n);
  YV4_REQUIRE(nseg >= 1 && nseg <= kMaxSeg, "sgd_step: 1 <= nseg <= %d (got %d)", kMaxSeg, nseg);
  YV4_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf | (uintptr_t)seg_hyper) & 15) == 0,
              "sgd_step: arenas and the hyper table must be 16-byte aligned");
  if (n == 0) return YV4_OK;
  hipLaunchKernelGGL(sgd_step_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     momentum_buf, n / 4, seg_off, reinterpret_cast<const float4*>(seg_hyper), nseg, ctrl);
  YV4_CHECK_LAUNCH("sgd_step");
  return YV4_OK;
}

int yv4_ema_update(float* ema, const float* online, int64_t n, float momentum, void* stream) {
  YV4_REQUIRE(ema && online, "ema_update: null pointer");
  YV4_REQUIRE(n >= 0 && n % 4 == 0, "ema_update: n must be a non-negative multiple of 4 (got %lld)", (long long)n);
  YV4_REQUIRE((((uintptr_t)ema | (uintptr_t)online) & 15) == 0, "ema_update: arenas must be 16-byte aligned");
  if (n == 0) return YV4_OK;
  // python: `1 - momentum` in double, then both become fp32 scalars of mul_/add_(alpha=)
  const float one_minus = (float)(1.0 - (double)momentum);
  hipLaunchKernelGGL(ema_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, ema, online, n / 4, momentum,
                     one_minus);
  YV4_CHECK_LAUNCH("ema_update");
  return YV4_OK;
}

}  // extern "C"
