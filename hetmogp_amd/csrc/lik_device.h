// lik_device.h -- device-side variational expectations E_q(f)[log p(y|f)] and their derivatives with respect to
// the mean / variance of q(f), for the eight likelihoods of /root/reference/likelihoods/*.py (SURVEY.md 8a, rows
// L1-L8).  Results reproduce the reference's formulas including its clips and quirks:
//   Q1  Gamma / Beta: Gauss-Hermite weights divided by sqrt(pi) twice (gamma.py:110,139-141; beta.py:113,142-144)
//   Q2  Categorical: d/dm is the constant onehot(y)[d] - 1 (categorical.py:102-113)
// Lane mapping: closed forms, 1-D quadratures and Gamma (separable in its two functions) use ONE lane per row;
// Beta (100 nodes) and Categorical (10^(K-1) nodes) use ONE WAVE per row, nodes strided over the 64 lanes and
// reduced with wavefront shuffles.
#pragma once
#include "common.h"
#include "gh_tables.h"
#include "../../include/hetmogp_hip.h"

#include "rowpass.h"  // HMOGP_MAXJ / HMOGP_MAXQ

#define LIM_VAL 709.782712893384      // log(DBL_MAX): GPy safe_exp clip
#define SQRT_DBL_MAX 1.3407807929942596e154  // GPy safe_square clip
#define INV_SQRT_PI 0.5641895835477563

struct LikOut {
  double ve;
  double gm[HMOGP_MAXJ];
  double gv[HMOGP_MAXJ];
};

__device__ __forceinline__ double safe_exp(double f) { return exp(fmin(f, LIM_VAL)); }
__device__ __forceinline__ double safe_square(double f) {
  const double g = fmin(f, SQRT_DBL_MAX);
  return g * g;
}
__device__ __forceinline__ double clip(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// digamma for x > 0: upward recurrence to x >= 10, then the asymptotic series in Bernoulli numbers.
__device__ __forceinline__ double digamma_pos(double x) {
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double z = 1.0 / (x * x);
  const double y =
      z * (8.33333333333333333333e-2 +
           z * (-8.33333333333333333333e-3 +
                z * (3.96825396825396825397e-3 +
                     z * (-4.16666666666666666667e-3 +
                          z * (7.57575757575757575758e-3 + z * (-2.10927960927960927961e-2 + z * 8.33333333333333333333e-2))))));
  return log(x) - 0.5 / x - y + r;
}

// trigamma = Hurwitz zeta(2, x) for x > 0 (scipy.special.zeta(2, a) in gamma.py:98, beta.py:99-101).
__device__ __forceinline__ double trigamma_pos(double x) {
  double r = 0.0;
  while (x < 12.0) {
    r += 1.0 / (x * x);
    x += 1.0;
  }
  const double ix = 1.0 / x, z = ix * ix;
  const double s = 1.0 / 6.0 -
                   z * (1.0 / 30.0 -
                        z * (1.0 / 42.0 - z * (1.0 / 30.0 - z * (5.0 / 66.0 - z * (691.0 / 2730.0 - z * (7.0 / 6.0))))));
  return r + ix + 0.5 * z + ix * z * s;
}

// ------------------------------------------------------------------------------------------- closed forms
// gaussian.py:41-62
__device__ __forceinline__ void lik_gaussian(double y, double m, double v, double sigma, LikOut& o) {
  const double s2 = sigma * sigma;
  o.ve = -0.5 * log(2.0 * M_PI) - 0.5 * log(s2) - 0.5 * (y * y + m * m + v - 2.0 * m * y) / s2;
  o.gm[0] = -(m - y) / s2;
  o.gv[0] = -0.5 * (1.0 / s2);
}

// hetgaussian.py:46-73
__device__ __forceinline__ void lik_hetgaussian(double y, const double* m, const double* v, LikOut& o) {
  const double prec = clip(safe_exp(-m[1] + 0.5 * v[1]), -1e9, 1e9);
  const double sq = clip(safe_square(y) + safe_square(m[0]) + v[0] - 2.0 * m[0] * y, -1e9, 1e9);
  o.ve = -(0.5 * log(2.0 * M_PI)) - 0.5 * m[1] - 0.5 * prec * sq;
  o.gm[0] = prec * (y - m[0]);
  o.gm[1] = 0.5 * (prec * sq - 1.0);
  o.gv[0] = -0.5 * prec;
  o.gv[1] = -0.25 * prec * sq;
}

// ------------------------------------------------------------------------------------------- 1-D, T = 20
// bernoulli.py:31-36,66-111 ; poisson.py:31-34,56-95 ; exponential.py:28-32,58-99
template <int LIK>
__device__ __forceinline__ void lik_quad1d(double y, double yaux, double m, double v, LikOut& o) {
  const double s = sqrt(2.0 * v);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 4
  for (int i = 0; i < 20; ++i) {
    const double f = GH20_X[i] * s + m, w = GH20_WN[i];
    double lp, d1, d2;
    if (LIK == HMOGP_LIK_BERNOULLI) {
      const double ef = safe_exp(f);
      const double p = clip(ef / (1.0 + ef), 1e-9, 1.0 - 1e-9);
      lp = y * log(p) + (1.0 - y) * log(1.0 - p);
      d1 = ((y - p) / (1.0 - p)) * (1.0 / (1.0 + ef));
      d2 = -p / (1.0 + ef);
    } else if (LIK == HMOGP_LIK_POISSON) {
      const double ef = safe_exp(f);
      lp = -ef + y * f - yaux;  // yaux = gammaln(y + 1)
      d1 = -ef + y;
      d2 = -ef;
    } else {  // exponential
      const double b = clip(safe_exp(-f), 1e-9, 1e9);
      lp = -log(b) - y / b;
      d1 = 1.0 - y / b;
      d2 = -y / b;
    }
    a0 += lp * w;
    a1 += d1 * w;
    a2 += d2 * w;
  }
  o.ve = a0;
  o.gm[0] = a1;
  o.gv[0] = 0.5 * a2;
}

// ------------------------------------------------------------------------------------------- Gamma, 10 x 10
// gamma.py:34-41,80-194.  a = exp(f1) depends only on node i, b = exp(f2) only on node j, and every term of
// logp / dlogp / d2logp is a product of a function of i and a function of j, so the 100-node tensor rule
// collapses to 10 + 10 evaluations.  Effective per-dimension weight w_i/pi (quirk Q1).
__device__ __forceinline__ void lik_gamma(double y, const double* m, const double* v, LikOut& o) {
  const double s1 = sqrt(2.0 * v[0]), s2 = sqrt(2.0 * v[1]);
  double S0 = 0.0, A1 = 0.0, Alg = 0.0, Apsi = 0.0, Azeta = 0.0, B1 = 0.0, Blog = 0.0;
  for (int i = 0; i < 10; ++i) {
    const double w = GH10_WN[i] * INV_SQRT_PI;
    const double a = clip(safe_exp(GH10_X[i] * s1 + m[0]), 1e-9, 1e9);
    const double b = clip(safe_exp(GH10_X[i] * s2 + m[1]), 1e-9, 1e9);
    S0 += w;
    A1 += w * a;
    Alg += w * lgamma(a);
    Apsi += w * digamma_pos(a) * a;
    Azeta += w * a * a * trigamma_pos(a);
    B1 += w * b;
    Blog += w * log(b);
  }
  const double ly = log(y);
  o.ve = -S0 * Alg + A1 * Blog + ly * S0 * (A1 - S0) - y * S0 * B1;
  const double common = A1 * Blog + ly * S0 * A1;
  o.gm[0] = -S0 * Apsi + common;
  o.gm[1] = S0 * A1 - y * S0 * B1;
  o.gv[0] = 0.5 * (-S0 * (Apsi + Azeta) + common);
  o.gv[1] = 0.5 * (-y * S0 * B1);
}

// Per-wave LDS scratch of the tensor-rule likelihoods (doubles): Categorical [0,80) exp(f_k(node i)), [80,160) f_k(node i),
// [160,170) normalised GH weights; Beta [0,80) a_i, psi(a_i), zeta(2,a_i), lgamma(a_i) and the same four for b_j.
#define HMOGP_ETAB 176

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// 1/d for a positive normal double: hardware seed + two Newton steps (error ~1e-16 relative; NOT correctly rounded)
__device__ __forceinline__ double fast_rcp_pos(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// log(x) for a positive normal double: x = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m-1)/(m+1),
// |s| <= 0.1716: odd series to s^21.  Absolute error ~2e-16 (1 + |log x| eps); about half the instructions of the
// library log (no special cases: the caller guarantees 1 <= x < 1e300).
__device__ __forceinline__ double fast_log_pos(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  const bool lowm = m < 0.70710678118654752;
  m = lowm ? m + m : m;
  e = lowm ? e - 1 : e;
  const double s = (m - 1.0) * fast_rcp_pos(m + 1.0), z = s * s;
  double p = 1.0 / 21.0;
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  p = fma(p * z, s + s, s + s);               // 2 s (1 + z p)
  const double de = (double)e;
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, p));
}

// ------------------------------------------------------------------------------------------- Beta, 10 x 10
// beta.py:29-36,76-197.  betaln / psi / zeta of (a+b) couple the two dimensions: 100 nodes over the 64 lanes.  Everything
// that depends on ONE dimension only -- a_i, psi(a_i), zeta(2, a_i), lgamma(a_i) and the same for b_j -- is evaluated once
// per row by 20 lanes into the wave's LDS table: 3 special functions per node (of a+b) instead of 9.
__device__ __forceinline__ void lik_beta_wave(double y, const double* m, const double* v, int lane, double* tab, LikOut& o) {
  if (lane < 20) {
    const int dim = lane / 10, i = lane - 10 * dim;
    const double x = clip(safe_exp(GH10_X[i] * sqrt(2.0 * v[dim]) + m[dim]), 1e-9, 1e9);
    double* t = tab + dim * 40 + i;
    t[0] = x, t[10] = digamma_pos(x), t[20] = trigamma_pos(x), t[30] = lgamma(x);
  }
  __builtin_amdgcn_wave_barrier();  // written and read by this wave only (LDS operations of a wave are in order)
  const double ly = log(y), l1y = log(1.0 - y);
  double ve = 0.0, g0 = 0.0, g1 = 0.0, h0 = 0.0, h1 = 0.0;
  for (int n = lane; n < 100; n += 64) {
    const int i = n / 10, j = n - 10 * i;
    const double w = (GH10_WN[i] * INV_SQRT_PI) * (GH10_WN[j] * INV_SQRT_PI);
    const double a = tab[i], pa = tab[10 + i], za = tab[20 + i], lga = tab[30 + i];
    const double b = tab[40 + j], pb = tab[50 + j], zb = tab[60 + j], lgb = tab[70 + j];
    const double pab = digamma_pos(a + b), zab = trigamma_pos(a + b);
    const double lbeta = lga + lgb - lgamma(a + b);
    ve += w * ((a - 1.0) * ly + (b - 1.0) * l1y - lbeta);
    g0 += w * ((pab - pa + ly) * a);
    g1 += w * ((pab - pb + l1y) * b);
    h0 += w * ((pab + a * zab - pa - a * za + ly) * a);
    h1 += w * ((pab + b * zab - pb - b * zb + l1y) * b);
  }
  o.ve = wave_sum(ve);
  o.gm[0] = wave_sum(g0);
  o.gm[1] = wave_sum(g1);
  o.gv[0] = 0.5 * wave_sum(h0);
  o.gv[1] = 0.5 * wave_sum(h1);
}

// ------------------------------------------------------------------------------------------- Categorical
// categorical.py:37-46,77-82,102-222.  K classes, D = K-1 functions, labels 1..K (class K = reference class),
// 10^D tensor nodes.  The node dependence factorises through exp(f_k) per dimension: a per-wave LDS
// table holds exp(f_k(node i)) and f_k(node i), [D][10] each.  The LAST min(D,3) dimensions are strided over the lanes
// (their digits, table entries and weight product are formed once per lane and chunk); the remaining leading dimensions
// are a uniform outer loop.  A node whose probabilities are not touched by the reference's clip to [1e-9, 1-1e-9] takes
//   log p_y = f_y - log(den),  d2 = -p_d (den - e_d)/den      (one reciprocal, one log per node)
// which equals the reference's clipped / renormalised expressions to rounding; any other node (clip active, overflow)
// takes the literal formulas.
template <int D>
__device__ __forceinline__ void cat_node_literal(const double (&e)[D], double w, int label, bool exact_dm, double& ve,
                                                 double (&hv)[D], double (&gx)[D]) {
  constexpr int K = D + 1;
  double esum = 0.0;
#pragma unroll
  for (int k = D - 1; k >= 0; --k) esum += e[k];
  const double den = 1.0 + esum;
  double psum = 0.0, py = 0.0;  // class probabilities, clipped then renormalised (:41-44)
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double pk = clip(e[k] / den, 1e-9, 1.0 - 1e-9);
    psum += pk;
    if (label == k + 1) py = pk;
  }
  const double pK = clip(1.0 / den, 1e-9, 1.0 - 1e-9);
  psum += pK;
  if (label == K) py = pK;
  ve += w * log(py / psum);
  // second derivative of log p wrt f_d (:115-128): -(e_d + sum_{j != d} e^{f_j + f_d}) / den^2, independent of y
  const double den2 = safe_square(den);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    double num = e[d];
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (j != d) num += fmin(e[j] * e[d], 1.79769313486231570815e308);
    hv[d] += w * (num / den2);
    if (exact_dm) gx[d] += w * ((label == d + 1 ? 1.0 : 0.0) - e[d] / den);  // E[d log p_y / d f_d], softmax
  }
}

// One node whose probabilities ARE touched by the clip, denominators far from overflow (den < 1e150): the literal formulas
// with every division replaced by a multiplication with a Newton-refined reciprocal and the library log by fast_log_pos
// (<= 2 ulp away from the reference's IEEE divisions; a clip comparison can only flip for a value within an ulp of the bound).
template <int D>
__device__ __forceinline__ void cat_node_clipped(const double (&e)[D], double den, double w, int label, bool exact_dm,
                                                 double& ve, double (&hv)[D], double (&gx)[D]) {
  constexpr int K = D + 1;
  const double rden = fast_rcp_pos(den);
  double psum = 0.0, py = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double pk = clip(e[k] * rden, 1e-9, 1.0 - 1e-9);
    psum += pk;
    if (label == k + 1) py = pk;
  }
  const double pK = clip(rden, 1e-9, 1.0 - 1e-9);
  psum += pK;
  if (label == K) py = pK;
  ve = fma(w, fast_log_pos(py) - fast_log_pos(psum), ve);
#pragma unroll
  for (int d = 0; d < D; ++d) {   // (e_d + sum_{j != d} e_j e_d) / den^2 = e_d (den - e_d) / den^2, no overflow below 1e150
    const double pd = e[d] * rden;
    hv[d] = fma(w * pd, (den - e[d]) * rden, hv[d]);
    if (exact_dm) gx[d] = fma(w, (label == d + 1 ? 1.0 : 0.0) - pd, gx[d]);
  }
}

// One node on the "clip inactive" path: log p_y = f_y - log(den), d2 log p / df_d^2 = -p_d (den - e_d) / den.
template <int D>
__device__ __forceinline__ void cat_node_fast(const double (&e)[D], double den, double w, double fy, int label, bool exact_dm,
                                              double& ve, double (&hv)[D], double (&gx)[D]) {
  const double rden = fast_rcp_pos(den);
  ve = fma(w, fy - fast_log_pos(den), ve);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double pd = e[d] * rden;
    hv[d] = fma(w * pd, (den - e[d]) * rden, hv[d]);
    if (exact_dm) gx[d] = fma(w, (label == d + 1 ? 1.0 : 0.0) - pd, gx[d]);
  }
}

// Loop structure (D functions, nodes in the reference's C order, function 0 slowest):
//   * the LAST min(D,3) dimensions are strided over the 64 lanes in chunks; their digits, table entries, weight product
//     and partial sum are formed once per lane and chunk;
//   * when D > 3, dimension D-4 is the REGISTER dimension: its ten table entries (exp f, f, weight) are preloaded into
//     registers once per row and its loop is fully unrolled -- ten independent nodes per trip, no LDS access and no index
//     arithmetic inside, so the scheduler can interleave their reciprocal / logarithm chains;
//   * dimensions 0 .. D-5 (D > 4) form a uniform outer loop whose table reads are amortised over those ten nodes.
// Whether the ten nodes of a trip may all take the fast path is decided once per trip from the largest / smallest
// denominator of the trip; otherwise each node of the trip is routed individually.
template <int D>
__device__ __forceinline__ void lik_categorical_t(double y, const double* m, const double* v, int lane, double* tab,
                                                  unsigned quirks, LikOut& o) {
  constexpr int K = D + 1;
  constexpr int DI = D < 3 ? D : 3;                          // lane-strided inner dimensions: functions D-DI .. D-1
  constexpr int HASR = D > 3 ? 1 : 0;                        // register dimension: function D-4
  constexpr int DS = D - DI - HASR;                          // slow (uniform) outer dimensions: functions 0 .. DS-1
  constexpr int NIN = DI == 1 ? 10 : (DI == 2 ? 100 : 1000);
  constexpr int RD = D - DI - 1;                             // index of the register dimension (valid when HASR)
  int NSLOW = 1;
#pragma unroll
  for (int k = 0; k < DS; ++k) NSLOW *= 10;
  double* etab = tab;
  double* ftab = tab + 80;
  double* wtab = tab + 160;
  for (int t = lane; t < D * 10; t += 64) {
    const int k = t / 10, i = t - 10 * k;
    const double f = GH10_X[i] * sqrt(2.0 * v[k]) + m[k];
    etab[t] = safe_exp(f), ftab[t] = f;
  }
  if (lane < 10) wtab[lane] = GH10_WN[lane];
  __builtin_amdgcn_wave_barrier();  // the tables are written and read by this wave only (LDS ops are in order)
  const int label = (int)y;  // 1..K
  const bool valid = (y == (double)label) && label >= 1 && label <= K;
  const bool exact_dm = (quirks & HMOGP_QUIRK_CATEGORICAL_DM) == 0;
  double er[HASR ? 10 : 1], wr[HASR ? 10 : 1], fr[HASR ? 10 : 1];
  if (HASR) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      er[i] = etab[RD * 10 + i], wr[i] = wtab[i];
      fr[i] = (label == RD + 1) ? ftab[RD * 10 + i] : 0.0;
    }
  }
  double ve = 0.0, hv[D], gx[D];
#pragma unroll
  for (int k = 0; k < D; ++k) hv[k] = gx[k] = 0.0;
  for (int base = 0; base < NIN; base += 64) {
    const int idx = base + lane;
    if (idx >= NIN) continue;
    // A node takes the fast path when the reference's clip to [1e-9, 1-1e-9] cannot touch any of its K probabilities
    // t / den, t in {e_0 .. e_{D-1}, 1}:  min t >= 1e-9 den  and  max t <= (1 - 1e-9) den  (and den finite).  The smallest /
    // largest t of the inner and slow dimensions are hoisted out of the register dimension's loop.
    double e[D], fy_in = 0.0, win = 1.0, sin_ = 0.0, lo_in = 1.0, hi_in = 1.0;
    int rem = idx;
#pragma unroll
    for (int t = DI - 1; t >= 0; --t) {  // digits of the inner dimensions, fastest first (the reference's weight order)
      const int i = rem % 10;
      rem /= 10;
      const int k = D - DI + t;
      e[k] = etab[k * 10 + i];
      if (label == k + 1) fy_in = ftab[k * 10 + i];
      win *= wtab[i];
      sin_ += e[k];
      lo_in = fmin(lo_in, e[k]), hi_in = fmax(hi_in, e[k]);
    }
    for (int sc = 0; sc < NSLOW; ++sc) {
      double ws = win, fys = fy_in, ss = sin_, lo = lo_in, hi = hi_in;
      int r2 = sc;
#pragma unroll
      for (int k = DS - 1; k >= 0; --k) {  // uniform digits of the slow outer dimensions
        const int i = r2 % 10;
        r2 /= 10;
        e[k] = etab[k * 10 + i];
        if (label == k + 1) fys = ftab[k * 10 + i];
        ws *= wtab[i];
        ss += e[k];
        lo = fmin(lo, e[k]), hi = fmax(hi, e[k]);
      }
      if (!HASR) {
        const double den = 1.0 + ss;
        if (den < 1e300 && den * 1e-9 <= lo && den * (1.0 - 1e-9) >= hi)
          cat_node_fast<D>(e, den, ws, fys, label, exact_dm, ve, hv, gx);
        else if (den < 1e150)
          cat_node_clipped<D>(e, den, ws, label, exact_dm, ve, hv, gx);
        else
          cat_node_literal<D>(e, ws, label, exact_dm, ve, hv, gx);
      } else {
        // NB the weight of the register dimension multiplies BEFORE the slow dimensions' in the reference's order; the
        // product of ten normalised weights is formed here as (inner * register) * slow -- same factors, and the fast
        // path only promises 1e-15 anyway; the literal path below restores the reference's order exactly.
        bool all_fast = true;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          const double den = 1.0 + (ss + er[i]);
          all_fast = all_fast && den < 1e300 && den * 1e-9 <= fmin(lo, er[i]) && den * (1.0 - 1e-9) >= fmax(hi, er[i]);
        }
        if (all_fast) {
          // Ten fast-path nodes that differ in the register dimension only.  For every OTHER dimension d the entry e_d is the
          // same in all ten, so  sum_i w_i p_di (1 - p_di) = e_d (A - e_d B)  with  A = sum_i w_i / den_i,
          // B = sum_i w_i / den_i^2  (and sum_i w_i (delta - p_di) = delta W - e_d A): two running sums per node instead
          // of five operations per node and dimension.  (Cancellation only where p_d -> 1, i.e. where the terms themselves
          // vanish against the total; the fast path is only entered for 1e-9 <= p <= 1 - 1e-9.)
          double A = 0.0, B = 0.0, Wt = 0.0;
#pragma unroll
          for (int i = 0; i < 10; ++i) {
            const double den = 1.0 + (ss + er[i]), w = ws * wr[i];
            const double rden = fast_rcp_pos(den);
            ve = fma(w, (fys + fr[i]) - fast_log_pos(den), ve);
            const double w1 = w * rden, w2 = w1 * rden;
            A += w1, B += w2;
            hv[RD] = fma(w2, er[i] * (den - er[i]), hv[RD]);
            if (exact_dm) {
              Wt += w;
              gx[RD] = fma(-w1, er[i], gx[RD]);
            }
          }
          if (exact_dm && label == RD + 1) gx[RD] += Wt;
#pragma unroll
          for (int d = 0; d < D; ++d) {
            if (d == RD) continue;
            hv[d] = fma(e[d], fma(-e[d], B, A), hv[d]);
            if (exact_dm) gx[d] += (label == d + 1 ? Wt : 0.0) - e[d] * A;
          }
        } else {
#pragma unroll 1
          for (int i = 0; i < 10; ++i) {
            e[RD] = etab[RD * 10 + i];
            double w = win * wtab[i];  // reference order: inner dims (fastest first), then RD, then the slow dims
            int r3 = sc;
#pragma unroll
            for (int k = DS - 1; k >= 0; --k) {
              w *= wtab[r3 % 10];
              r3 /= 10;
            }
            const double den = 1.0 + (ss + e[RD]);
            if (den < 1e300 && den * 1e-9 <= fmin(lo, e[RD]) && den * (1.0 - 1e-9) >= fmax(hi, e[RD]))
              cat_node_fast<D>(e, den, w, fys + ((label == RD + 1) ? ftab[RD * 10 + i] : 0.0), label, exact_dm, ve, hv, gx);
            else if (den < 1e150)
              cat_node_clipped<D>(e, den, w, label, exact_dm, ve, hv, gx);
            else
              cat_node_literal<D>(e, w, label, exact_dm, ve, hv, gx);
          }
        }
      }
    }
  }
  o.ve = valid ? wave_sum(ve) : nan("");
  double wpow = 1.0;
#pragma unroll
  for (int k = 0; k < D; ++k) wpow *= GH10_WSUM_OVER_SQRTPI;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double s = wave_sum(hv[d]);
    o.gv[d] = valid ? -0.5 * s : 0.0;
    if (exact_dm) {
      const double g = wave_sum(gx[d]);
      o.gm[d] = valid ? g : 0.0;
    } else {
      o.gm[d] = ((label == d + 1 ? 1.0 : 0.0) - (valid ? 1.0 : 0.0)) * wpow;  // quirk Q2
    }
  }
}

// ============================================================================ predictive moments
// `<likelihood>.predictive(m, v)`: mean and variance of y under q(f) = N(m, diag v).  One lane per row for the closed
// forms and 1-D rules, one wave per row for the T x T (Gamma, Beta) and 10^(K-1) (Categorical) tensor rules.
// T = 20 on a fresh reference instance, 10 when var_exp ran first on it (GPy caches the first rule, quirk Q7).
__device__ __forceinline__ double gh_x(int T, int i) { return T == 10 ? GH10_X[i] : GH20_X[i]; }
__device__ __forceinline__ double gh_wn(int T, int i) { return T == 10 ? GH10_WN[i] : GH20_WN[i]; }

template <int LIK>
__device__ __forceinline__ void lik_predictive(const double* m, const double* v, double param, int T, int lane, double* etab,
                                               double* mean, double* var) {
  if (LIK == HMOGP_LIK_GAUSSIAN) {  // gaussian.py:64-67
    mean[0] = m[0];
    var[0] = param * param + v[0];
  } else if (LIK == HMOGP_LIK_HETGAUSSIAN) {  // hetgaussian.py:75-88
    const double s1 = sqrt(2.0 * v[0]), s2 = sqrt(2.0 * v[1]);
    double e2 = 0.0, sq = 0.0;
    for (int i = 0; i < T; ++i) {
      const double w = gh_wn(T, i);
      e2 += safe_exp(gh_x(T, i) * s2 + m[1]) * w;
      sq += safe_square(gh_x(T, i) * s1 + m[0]) * w;
    }
    mean[0] = m[0];
    var[0] = e2 + sq - m[0] * m[0];
  } else if (LIK == HMOGP_LIK_BERNOULLI || LIK == HMOGP_LIK_POISSON || LIK == HMOGP_LIK_EXPONENTIAL) {
    const double s = sqrt(2.0 * v[0]);  // bernoulli.py:113-128, poisson.py:97-112, exponential.py:101-116
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = 0; i < T; ++i) {
      const double f = gh_x(T, i) * s + m[0], w = gh_wn(T, i);
      double mu, vr, ms;
      if (LIK == HMOGP_LIK_BERNOULLI) {
        const double ef = safe_exp(f);
        const double p = clip(ef / (1.0 + ef), 1e-9, 1.0 - 1e-9);
        mu = p, vr = p * (1.0 - p), ms = p * p;
      } else if (LIK == HMOGP_LIK_POISSON) {
        const double ef = safe_exp(f);
        mu = ef, vr = ef, ms = ef * ef;
      } else {
        const double b = clip(safe_exp(-f), 1e-9, 1e9);
        mu = b, vr = safe_square(b), ms = safe_square(b);
      }
      a0 += mu * w;
      a1 += vr * w;
      a2 += ms * w;
    }
    mean[0] = a0;
    var[0] = a1 + a2 - a0 * a0;
  } else if (LIK == HMOGP_LIK_GAMMA || LIK == HMOGP_LIK_BETA) {  // gamma.py:196-238, beta.py:199-241 (quirk Q1)
    const double s1 = sqrt(2.0 * v[0]), s2 = sqrt(2.0 * v[1]);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int n = lane; n < T * T; n += 64) {
      const int i = n / T, j = n - T * i;
      const double w = (gh_wn(T, i) * INV_SQRT_PI) * (gh_wn(T, j) * INV_SQRT_PI);
      const double a = clip(safe_exp(gh_x(T, i) * s1 + m[0]), 1e-9, 1e9);
      const double b = clip(safe_exp(gh_x(T, j) * s2 + m[1]), 1e-9, 1e9);
      double mu, vr;
      if (LIK == HMOGP_LIK_GAMMA) {
        mu = a / b;
        vr = a / (b * b);
      } else {
        mu = a / (a + b);
        vr = a * b / ((a + b) * (a + b) * (a + b + 1.0));
      }
      a0 += w * mu;
      a1 += w * vr;
      a2 += w * mu * mu;
    }
    a0 = wave_sum(a0), a1 = wave_sum(a1), a2 = wave_sum(a2);
    mean[0] = a0;
    var[0] = a1 + a2 - safe_square(a0);
  } else {  // Categorical, categorical.py:84-99,224-269: E[rho_d], rho normalised over the K-1 columns; variance zeros
    const int D = (int)param - 1;
    for (int e = lane; e < D * 10; e += 64) {
      const int k = e / 10, i = e - 10 * k;
      etab[e] = safe_exp(GH10_X[i] * sqrt(2.0 * v[k]) + m[k]);
    }
    __builtin_amdgcn_wave_barrier();
    int total = 1;
    for (int k = 0; k < D; ++k) total *= 10;
    double acc[HMOGP_MAXJ];
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k) acc[k] = 0.0;
    for (int n = lane; n < total; n += 64) {
      double e[HMOGP_MAXJ];
      double w = 1.0, esum = 0.0;
      int rem = n;
#pragma unroll
      for (int k = HMOGP_MAXJ - 1; k >= 0; --k) {
        if (k < D) {
          const int i = rem % 10;
          rem /= 10;
          e[k] = etab[k * 10 + i];
          w *= GH10_WN[i];
          esum += e[k];
        } else {
          e[k] = 0.0;
        }
      }
      double rs = 0.0;
#pragma unroll
      for (int k = 0; k < HMOGP_MAXJ; ++k)
        if (k < D) {
          e[k] = clip(e[k] / (1.0 + esum), 1e-9, 1.0 - 1e-9);
          rs += e[k];
        }
#pragma unroll
      for (int k = 0; k < HMOGP_MAXJ; ++k)
        if (k < D) acc[k] += w * (e[k] / rs);
    }
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k)
      if (k < D) {
        mean[k] = wave_sum(acc[k]);
        var[k] = 0.0;
      }
  }
}

// ============================================================================ Monte-Carlo log predictive
// log p(y|f) at ONE sample f of q(f), as the reference's `log_predictive` evaluates it (gaussian.py:28-34 -- sigma is
// ignored, quirk Q6 --, bernoulli.py:31-36, hetgaussian.py:35-39, poisson.py:31-34, exponential.py:28-32,
// categorical.py:48-63).  Gamma and Beta have no log_predictive in the reference.
template <int LIK>
__device__ __forceinline__ double lik_logpdf_sample(double y, double yaux, const double* f, double param) {
  if (LIK == HMOGP_LIK_GAUSSIAN) {
    const double d = y - f[0];
    return -0.5 * log(2.0 * M_PI) - 0.5 * d * d;
  } else if (LIK == HMOGP_LIK_BERNOULLI) {
    const double ef = safe_exp(f[0]);
    const double p = clip(ef / (1.0 + ef), 1e-9, 1.0 - 1e-9);
    return y * log(p) + (1.0 - y) * log(1.0 - p);
  } else if (LIK == HMOGP_LIK_HETGAUSSIAN) {
    const double ev = safe_exp(f[1]);
    return -0.5 * log(2.0 * M_PI) - 0.5 * f[1] - 0.5 * (safe_square(y - f[0]) / ev);
  } else if (LIK == HMOGP_LIK_POISSON) {
    return -safe_exp(f[0]) + y * f[0] - yaux;
  } else if (LIK == HMOGP_LIK_EXPONENTIAL) {
    const double b = clip(safe_exp(-f[0]), 1e-9, 1e9);
    return -log(b) - y / b;
  } else if (LIK == HMOGP_LIK_CATEGORICAL) {
    const int K = (int)param, D = K - 1, label = (int)y;
    double esum = 0.0, e[HMOGP_MAXJ];
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k) {
      e[k] = (k < D) ? safe_exp(f[k]) : 0.0;
      esum += e[k];
    }
    const double den = 1.0 + esum;
    double psum = 0.0, py = 0.0;
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k)
      if (k < D) {
        const double pk = clip(e[k] / den, 1e-9, 1.0 - 1e-9);
        psum += pk;
        if (label == k + 1) py = pk;
      }
    const double pK = clip(1.0 / den, 1e-9, 1.0 - 1e-9);
    psum += pK;
    if (label == K) py = pK;
    return log(py / psum);
  }
  return nan("");
}

// counter-based generator: two standard normals from (seed, row, sample, pair) via splitmix64 + Box-Muller
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
__device__ __forceinline__ void normal_pair(unsigned long long seed, long long n, int s, int pair, double& z0, double& z1) {
  const unsigned long long h = splitmix64(splitmix64(seed ^ (unsigned long long)n * 0xD1342543DE82EF95ULL) ^
                                          ((unsigned long long)s << 8) ^ (unsigned long long)pair);
  const unsigned long long h2 = splitmix64(h);
  const double u1 = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0, 1)
  const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);         // [0, 1)
  const double r = sqrt(-2.0 * log(u1));
  z0 = r * cos(2.0 * M_PI * u2);
  z1 = r * sin(2.0 * M_PI * u2);
}

// ============================================================================ data generation (`samples`)
// One draw y ~ p(y | f) per row with the link functions and clips of the reference's `<likelihood>.samples`
// (gaussian.py:36-39, bernoulli.py:59-64, hetgaussian.py:41-44, poisson.py:51-54, exponential.py:52-56, gamma.py:43-50,
// beta.py:38-45, categorical.py:65-75).  Counter-based generator (reproducible per (seed, row); a different stream than
// NumPy's, so only the DISTRIBUTION is comparable with the reference): uniforms from splitmix64, normals by Box-Muller,
// Poisson by multiplication (lambda < 10) / Hoermann's PTRS transformed rejection, Gamma by Marsaglia-Tsang.
struct RowRng {
  unsigned long long key, ctr;
  __device__ RowRng(unsigned long long seed, long long row) : key(splitmix64(seed ^ (unsigned long long)row * 0xD1342543DE82EF95ULL)), ctr(0) {}
  __device__ double uniform() {  // (0, 1)
    const unsigned long long h = splitmix64(key ^ (++ctr * 0x9E3779B97F4A7C15ULL));
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  }
  __device__ double normal() {
    const double u1 = uniform(), u2 = uniform();
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
  }
  __device__ double gamma(double a) {  // shape a > 0, scale 1
    double boost = 1.0;
    if (a < 1.0) {
      boost = pow(uniform(), 1.0 / a);
      a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 1000; ++it) {
      const double x = normal(), t = 1.0 + c * x;
      if (t <= 0.0) continue;
      const double vv = t * t * t, u = uniform();
      if (u < 1.0 - 0.0331 * (x * x) * (x * x) || log(u) < 0.5 * x * x + d * (1.0 - vv + log(vv))) return boost * d * vv;
    }
    return boost * d;
  }
  __device__ double poisson(double lam) {
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {
      const double L = exp(-lam);
      double k = 0.0, p = uniform();
      while (p > L) {
        k += 1.0;
        p *= uniform();
      }
      return k;
    }
    if (lam > 1e15) return lam;  // beyond float64 integer resolution the distribution is its mean
    const double slam = sqrt(lam), loglam = log(lam), b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 1000; ++it) {
      const double U = uniform() - 0.5, V = uniform(), us = 0.5 - fabs(U);
      const double k = floor((2.0 * a / us + b) * U + lam + 0.43);
      if (us >= 0.07 && V <= vr) return k;
      if (k < 0.0 || (us < 0.013 && V > us)) continue;
      if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
    }
    return floor(lam);
  }
};

template <int LIK>
__device__ __forceinline__ double lik_sample(RowRng& g, const double* f, double param) {
  if (LIK == HMOGP_LIK_GAUSSIAN) {
    return f[0] + param * g.normal();
  } else if (LIK == HMOGP_LIK_BERNOULLI) {
    const double ef = safe_exp(f[0]);
    return g.uniform() < clip(ef / (1.0 + ef), 1e-9, 1.0 - 1e-9) ? 1.0 : 0.0;
  } else if (LIK == HMOGP_LIK_HETGAUSSIAN) {
    return f[0] + sqrt(safe_exp(f[1])) * g.normal();
  } else if (LIK == HMOGP_LIK_POISSON) {
    return g.poisson(safe_exp(f[0]));
  } else if (LIK == HMOGP_LIK_EXPONENTIAL) {
    return -clip(safe_exp(-f[0]), 1e-9, 1e9) * log(g.uniform());
  } else if (LIK == HMOGP_LIK_GAMMA) {
    const double a = clip(safe_exp(f[0]), 1e-9, 1e9), b = clip(safe_exp(f[1]), 1e-9, 1e9);
    return g.gamma(a) / b;
  } else if (LIK == HMOGP_LIK_BETA) {
    const double a = clip(safe_exp(f[0]), 1e-9, 1e9), b = clip(safe_exp(f[1]), 1e-9, 1e9);
    const double x = g.gamma(a), yv = g.gamma(b);
    return x / (x + yv);
  } else {  // Categorical: labels 1..K, probabilities clipped then renormalised (categorical.py:66-71)
    const int K = (int)param, D = K - 1;
    double e[HMOGP_MAXJ], esum = 0.0;
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k) {
      e[k] = (k < D) ? safe_exp(f[k]) : 0.0;
      esum += e[k];
    }
    const double den = 1.0 + esum;
    double psum = clip(1.0 / den, 1e-9, 1.0 - 1e-9);
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k)
      if (k < D) {
        e[k] = clip(e[k] / den, 1e-9, 1.0 - 1e-9);
        psum += e[k];
      }
    const double u = g.uniform() * psum;
    double cum = 0.0, label = (double)K;
    bool found = false;
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k)
      if (k < D && !found) {
        cum += e[k];
        if (u < cum) label = (double)(k + 1), found = true;
      }
    return label;
  }
}

// lanes per row of a likelihood's predictive rule
__host__ __device__ constexpr int lik_pred_lanes(int lik) {
  return (lik == HMOGP_LIK_BETA || lik == HMOGP_LIK_GAMMA || lik == HMOGP_LIK_CATEGORICAL) ? 64 : 1;
}

// lanes per row of a likelihood
__host__ __device__ constexpr int lik_lanes(int lik) {
  return (lik == HMOGP_LIK_BETA || lik == HMOGP_LIK_CATEGORICAL) ? 64 : 1;
}

// Dispatch.  For 64-lane likelihoods every lane of the wave must call with the same row; the result is valid in
// every lane.  `etab` (per-wave LDS, HMOGP_MAXJ*10 doubles) is only used by Categorical.
// CATD: number of functions (K-1) of a Categorical likelihood -- a template parameter so that each K gets its own register
// allocation (0 for every other likelihood).
template <int LIK, int CATD = 0>
__device__ __forceinline__ void lik_eval(double y, double yaux, const double* m, const double* v, double param, int lane,
                                         double* etab, unsigned quirks, LikOut& o) {
  if (LIK == HMOGP_LIK_GAUSSIAN)
    lik_gaussian(y, m[0], v[0], param, o);
  else if (LIK == HMOGP_LIK_HETGAUSSIAN)
    lik_hetgaussian(y, m, v, o);
  else if (LIK == HMOGP_LIK_BERNOULLI || LIK == HMOGP_LIK_POISSON || LIK == HMOGP_LIK_EXPONENTIAL)
    lik_quad1d<LIK>(y, yaux, m[0], v[0], o);
  else if (LIK == HMOGP_LIK_GAMMA)
    lik_gamma(y, m, v, o);
  else if (LIK == HMOGP_LIK_BETA)
    lik_beta_wave(y, m, v, lane, etab, o);
  else
    lik_categorical_t<(CATD > 0 ? CATD : 1)>(y, m, v, lane, etab, quirks, o);
  if ((LIK == HMOGP_LIK_GAMMA || LIK == HMOGP_LIK_BETA) && !(quirks & HMOGP_QUIRK_GAMMA_BETA_PI)) {
    o.ve *= M_PI;  // exact mode: undo the second division of each dimension's weights by sqrt(pi) (quirk Q1)
#pragma unroll
    for (int j = 0; j < 2; ++j) o.gm[j] *= M_PI, o.gv[j] *= M_PI;
  }
}
