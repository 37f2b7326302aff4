// Shared helpers of the gfx950 YOLOv4 kernels (device math + host error plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "yv4.h"

// Measurement switches.  The PRODUCT library (lib/libyv4_hip.so) reads no environment variable: every tuning knob is
// its measured default and the ablation branches (which compute wrong results on purpose, to time a kernel without one
// of its parts) do not exist.  `make measure` builds lib_alt/libyv4_hip_measure.so with -DYV4_MEASURE, in which
// YV4_ENV_INT reads the variable once and YV4_ABLATE tests the flag word; YV4_LIB_PATH points the binding at it.
#ifdef YV4_MEASURE
#include <cstdlib>
#define YV4_ENV_INT(NAME, DEFAULT) ([] { const char* e_ = getenv(NAME); return e_ ? atoi(e_) : (DEFAULT); }())
#define YV4_ABLATE(FLAGS, BIT) ((FLAGS) & (BIT))
#else
#define YV4_ENV_INT(NAME, DEFAULT) (DEFAULT)
#define YV4_ABLATE(FLAGS, BIT) false
#endif

namespace yv4 {

// ---- host side ---------------------------------------------------------------
void set_error(const char* fmt, ...);

#define YV4_REQUIRE(cond, ...)              \
  do {                                      \
    if (!(cond)) {                          \
      ::yv4::set_error(__VA_ARGS__);        \
      return YV4_E_INVALID;                 \
    }                                       \
  } while (0)

#define YV4_CHECK_LAUNCH(what)                                               \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      ::yv4::set_error("%s: %s", what, hipGetErrorString(e__));              \
      return YV4_E_LAUNCH;                                                   \
    }                                                                        \
  } while (0)

// ---- division by a launch-invariant divisor: one mulhi + shift instead of the ~35-instruction runtime divide
// (the per-row (n, ho, wo) decomposition in the conv prologues: 8 divides per lane were longer than the whole
// MFMA phase of the K <= 256 layers).  Exact for 0 <= x < 2^31.
struct FastDiv { unsigned mul, shr; };
__host__ __device__ static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f = {0u, 0u};
  if (d > 1) {
    unsigned lg = 0;
    while ((1ull << lg) < d) ++lg;                       // ceil(log2 d)
    const unsigned pw = 31 + lg;
    f.mul = (unsigned)(((1ull << pw) + d - 1) / d);
    f.shr = pw - 32;
  }
  return f;
}
__device__ __forceinline__ int fd_div(int x, FastDiv f) {
  return f.mul ? (int)(__umulhi((unsigned)x, f.mul) >> f.shr) : x;
}
// bit (kh*KW + kw) set when input pixel (hi0 + kh, wi0 + kw) is inside the image
__device__ __forceinline__ unsigned long long tap_mask(int hi0, int wi0, int KH, int KW, int H, int W) {
  unsigned colm = 0u;
  for (int kw = 0; kw < KW; ++kw) colm |= ((unsigned)(wi0 + kw) < (unsigned)W ? 1u : 0u) << kw;
  unsigned long long mk = 0ull;
  for (int kh = 0; kh < KH; ++kh)
    if ((unsigned)(hi0 + kh) < (unsigned)H) mk |= (unsigned long long)colm << (kh * KW);
  return mk;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON ONE DEVICE: a process that launches on a second
// GPU must set it there too.  One LdsAttrOnce per (call site, kernel) remembers the devices already done (bit per
// device ordinal); thread-safe, and the status of hipFuncSetAttribute is checked.
struct LdsAttrOnce { std::atomic<unsigned long long> done{0ull}; };
static inline int ensure_dyn_lds(LdsAttrOnce& g, const void* fn, size_t bytes, const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("%s: hipGetDevice failed", who);
    return YV4_E_LAUNCH;
  }
  const unsigned long long bit = 1ull << (dev & 63);
  if (g.done.load(std::memory_order_acquire) & bit) return YV4_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: cannot reserve %zu bytes of dynamic LDS on device %d", who, bytes, dev);
    return YV4_E_LAUNCH;
  }
  g.done.fetch_or(bit, std::memory_order_release);
  return YV4_OK;
}

// ---- deterministic mode (yv4_set_deterministic) -------------------------------
// Every floating-point sum that meets in atomics (BatchNorm statistics from the conv epilogues, the BatchNorm
// backward's dbeta / dgamma, the loss sums, the positives' row gradients, the bias gradients, the SPP scatter) is
// order-dependent in its last bits, and a 110-layer network under batch statistics amplifies one ulp to percents
// of the loss within a hundred steps.  With the mode on, those accumulators are FIXED-POINT INTEGERS: an addend v
// becomes round(v * 2^SHIFT) split into two 64-bit words (hi = floor(t / 2^32), lo = t - hi * 2^32 in [0, 2^32)),
// both added with integer atomics.  Integer addition is associative and commutative, so the result is the same bits
// whatever the arrival order, the workgroup-to-CU assignment or the replica an addend lands in -- no second pass, no
// ordered reduction, the kernels keep their structure.  Range of an addend AND of the sum: |v| < 2^(94 - SHIFT),
// resolution 2^-SHIFT.  A non-finite or out-of-range addend sets bit 63 of the lo word (an idempotent OR: lo sums stay
// below 2^63 for < 2^31 addends, so no carry ever reaches the bit) and the value reads back as NaN; the hi words are
// summed with wrapping 64-bit adds, so fx_value also answers NaN when the decoded sum has left the range (|H| >= 2^62:
// a sum that large is one or two addends away from wrapping into a wrong finite value) -- overflow stays loud, per
// addend and in total, as in the double path.
bool deterministic();          // api.hip: the process-wide switch, read by the launchers
constexpr int kFxStat = 40;    // forward statistics, loss sums: |v| < 1.8e16, resolution 9.1e-13
constexpr int kFxGrad = 60;    // gradient sums (loss-scaled): addends and sums |v| < 1.7e10, resolution 8.7e-19
typedef unsigned long long u64_t;

template <int SHIFT>
__device__ __forceinline__ void fx_add(u64_t* hi, u64_t* lo, double v) {
  double t = v * __builtin_ldexp(1.0, SHIFT);
  if (!(__builtin_fabs(t) < 0x1p94)) {          // NaN, infinity, out of range
    atomicOr(lo, 1ull << 63);
    return;
  }
  t = __builtin_rint(t);
  const double h = __builtin_floor(t * 0x1p-32);
  const double l = t - h * 0x1p32;              // exact, in [0, 2^32)
  const long long hv = (long long)h;
  if (hv) atomicAdd(hi, (u64_t)hv);
  atomicAdd(lo, (u64_t)(long long)l);
}
// words -> value.  (hi, lo) may be sums of several accumulators' words (fx_fold).
template <int SHIFT>
__device__ __forceinline__ double fx_value(u64_t hi, u64_t lo) {
  if (lo >> 63) return __builtin_nan("");
  const long long H = (long long)hi + (long long)(lo >> 32);
  if (H >= (1ll << 62) || H <= -(1ll << 62)) return __builtin_nan("");     // the SUM is out of range (see above)
  return ((double)H * 0x1p32 + (double)(lo & 0xffffffffull)) * __builtin_ldexp(1.0, -SHIFT);
}
__device__ __forceinline__ void fx_fold(u64_t& hi, u64_t& lo, u64_t h2, u64_t l2) {
  hi += h2;
  lo = ((lo & ~(1ull << 63)) + (l2 & ~(1ull << 63))) | ((lo | l2) & (1ull << 63));
}

// BatchNorm statistics replicas of the conv epilogues: stats = [YV4_STATS_REPLICAS][sum (C) | sum of squares (C)] doubles.
// The launcher tags bit 0 of the pointer when the mode is on; replicas are then used in PAIRS (even: hi words, odd: lo
// words, same [sum | sum of squares] layout) -- the buffer keeps its size, bn_finalize_kernel decodes.
struct StatRep { double* p; int lo_off; };      // lo_off = 0: plain doubles
__device__ __forceinline__ StatRep stat_rep(double* stats, unsigned idx, int C) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(stats);
  StatRep r;
  if (a & 1ull) {
    r.p = reinterpret_cast<double*>(a & ~7ull) + (size_t)(idx & (YV4_STATS_REPLICAS - 2)) * 2 * C;
    r.lo_off = 2 * C;
  } else {
    r.p = stats + (size_t)(idx & (YV4_STATS_REPLICAS - 1)) * 2 * C;
    r.lo_off = 0;
  }
  return r;
}
__device__ __forceinline__ void stat_add(const StatRep& r, int i, float v) {
  if (r.lo_off) fx_add<kFxStat>(reinterpret_cast<u64_t*>(r.p) + i, reinterpret_cast<u64_t*>(r.p) + r.lo_off + i, (double)v);
  else atomicAdd(&r.p[i], (double)v);
}
static inline double* tag_stats(double* stats) {
  return deterministic() && stats ? reinterpret_cast<double*>(reinterpret_cast<unsigned long long>(stats) | 1ull) : stats;
}

// ---- device math -------------------------------------------------------------
// Mish, mmdet/ops/mish_cuda/src/mish.h:16-18:  x * tanh(x < 20 ? log1p(exp(x)) : x).
// tanh(log1p(e)) == (e*e + 2e) / (e*e + 2e + 2) exactly, which needs one exp and
// one divide and keeps full relative precision for very negative x.
__device__ __forceinline__ float mish_f32(float x) {
  if (x >= 20.f) return x;  // tanh(x) rounds to 1 in fp32 for x >= 9.02
  const float e = expf(x);
  const float n = e * (e + 2.f);
  return x * (n / (n + 2.f));
}

// Epilogue form: hardware exp2 and reciprocal (v_exp_f32 / v_rcp_f32, 1 ulp each) instead of the
// libm expf and the IEEE divide: ~10 VALU instructions instead of ~35 per output element, which
// matters for the K <= 256 layers whose epilogue is as long as their MFMA phase.  Absolute error
// vs mish_f32 is < 2e-6 over the whole range (tests/test_gpu_parity.py::test_conv_*), far inside
// the 1e-4 parity budget.
__device__ __forceinline__ float mish_fast_f32(float x) {
  // no contraction in here: whether  n + 2  became fma(e, e + 2, 2) used to depend on the loop around the call, so two
  // kernels with the same accumulator could round an output differently (the fp32 plans promise the same bits from
  // every tile kernel, tests/test_gpu_parity.py::test_conv1x1_ws_kernel_f32)
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
  const float n = e * (e + 2.f);
  const float y = x * n * __builtin_amdgcn_rcpf(n + 2.f);
  return x >= 20.f ? x : y;
}

// Two at a time for the epilogues that store 16-bit values: the multiplies and adds become packed fp32 instructions
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, two elements per lane and issue slot), leaving the two quarter-rate
// transcendentals per element as the bulk of the cost: ~46 issue cycles per element and wave instead of ~64 -- these
// epilogues are VALU-bound on the HBM-bound layers.  The x >= 20 branch is replaced by clamping the exponent
// (for x >= 20 the ratio n / (n + 2) rounds to 1): the result may differ from mish_fast_f32's by one fp32 ulp there,
// invisible after the rounding to 16 bits.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t mish_fast2_f32(f32x2_t x) {
  const f32x2_t xs = x * 1.44269504088896340736f;
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(fminf(xs.x, 28.853900817779268f));   // e^20
  e.y = __builtin_amdgcn_exp2f(fminf(xs.y, 28.853900817779268f));
  const f32x2_t n = e * (e + 2.f);
  const f32x2_t d = n + 2.f;
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return x * (n * r);
}
template <int N>
__device__ __forceinline__ void mish_fast_row(float (&v)[N]) {
  static_assert(N % 2 == 0, "pairs");
#ifdef YV4_SCALAR_MISH      // A/B build: the scalar form in the same places
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = mish_fast_f32(v[u]);
  return;
#endif
#pragma unroll
  for (int u = 0; u < N; u += 2) {
    f32x2_t a;
    a.x = v[u]; a.y = v[u + 1];
    a = mish_fast2_f32(a);
    v[u] = a.x; v[u + 1] = a.y;
  }
}

__device__ __forceinline__ float sigmoid_f32(float x) {
  return 1.f / (1.f + expf(-x));
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
#ifdef YV4_EXACT_MISH
    case YV4_ACT_MISH: return mish_f32(v);
#else
    case YV4_ACT_MISH: return mish_fast_f32(v);
#endif
    case YV4_ACT_LEAKY: return v >= 0.f ? v : v * slope;
    case YV4_ACT_SWISH: return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504088896340736f));
    default: return v;
  }
}

}  // namespace yv4
