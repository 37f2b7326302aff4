/*
 * oracle/mish_ref.c -- CPU ORACLE (test infrastructure, not the product).
 *
 * The scalar loops of /root/reference/mmdet/ops/mish_cuda/src/kernel/mish_cpu.cc:6-29 over
 * the math of src/mish.h:16-29 (THRESHOLD 20), for float and double.
 */
#include <math.h>
#include <stdint.h>

/* With scalar_t = float the reference's unqualified exp/log1p/tanh calls (mish.h:17,22-26,
 * inside namespace mish_cpu_kernel) bind to the DOUBLE overloads: the forward expression is
 * evaluated in double and rounded once; the backward rounds to float at each named
 * `const scalar_t` (verified bit for bit against oracle/_ref on tests/golden/mish.npz). */
static float fwd_f(float x) { return (float)((double)x * tanh(x < 20.f ? log1p(exp((double)x)) : (double)x)); }
static float bwd_f(float g, float x) {
  const float sp = (float)(x < 20.f ? log1p(exp((double)x)) : (double)x);
  const float grad_sp = (float)(1 - exp(-(double)sp));
  const float tsp = (float)tanh((double)sp);
  const float grad_tsp = (1 - tsp * tsp) * grad_sp;
  const float grad = x * grad_tsp + tsp;
  return g * grad;
}
static double fwd_d(double x) { return x * tanh(x < 20.0 ? log1p(exp(x)) : x); }
static double bwd_d(double g, double x) {
  const double sp = x < 20.0 ? log1p(exp(x)) : x;
  const double grad_sp = 1.0 - exp(-sp);
  const double tsp = tanh(sp);
  const double grad_tsp = (1.0 - tsp * tsp) * grad_sp;
  const double grad = x * grad_tsp + tsp;
  return g * grad;
}

void oracle_mish_fwd_f32(const float* in, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = fwd_f(in[i]); }
void oracle_mish_bwd_f32(const float* g, const float* in, float* gin, int64_t n) { for (int64_t i = 0; i < n; ++i) gin[i] = bwd_f(g[i], in[i]); }
void oracle_mish_fwd_f64(const double* in, double* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = fwd_d(in[i]); }
void oracle_mish_bwd_f64(const double* g, const double* in, double* gin, int64_t n) { for (int64_t i = 0; i < n; ++i) gin[i] = bwd_d(g[i], in[i]); }
