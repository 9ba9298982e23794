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

// ------------------------------------------------------------------------------------------- Beta, 10 x 10
// beta.py:29-36,76-197.  betaln / psi / zeta of (a+b) couple the two dimensions: 100 nodes over the 64 lanes.
__device__ __forceinline__ void lik_beta_wave(double y, const double* m, const double* v, int lane, LikOut& o) {
  const double s1 = sqrt(2.0 * v[0]), s2 = sqrt(2.0 * v[1]);
  const double ly = log(y), l1y = log(1.0 - y);
  double ve = 0.0, g0 = 0.0, g1 = 0.0, h0 = 0.0, h1 = 0.0;
  for (int n = lane; n < 100; n += 64) {
    const int i = n / 10, j = n - 10 * i;
    const double w = (GH10_WN[i] * INV_SQRT_PI) * (GH10_WN[j] * INV_SQRT_PI);
    const double a = clip(safe_exp(GH10_X[i] * s1 + m[0]), 1e-9, 1e9);
    const double b = clip(safe_exp(GH10_X[j] * s2 + m[1]), 1e-9, 1e9);
    const double pab = digamma_pos(a + b), pa = digamma_pos(a), pb = digamma_pos(b);
    const double zab = trigamma_pos(a + b), za = trigamma_pos(a), zb = trigamma_pos(b);
    const double lbeta = lgamma(a) + lgamma(b) - lgamma(a + b);
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
// 10^D tensor nodes strided over the wave.  `etab` is a per-wave LDS table [D][10] of exp(f_k(node i)).
__device__ __forceinline__ void lik_categorical_wave(double y, const double* m, const double* v, int K, int lane,
                                                     double* etab, unsigned quirks, LikOut& o) {
  const int D = K - 1;
  for (int e = lane; e < D * 10; e += 64) {
    const int k = e / 10, i = e - 10 * k;
    etab[e] = safe_exp(GH10_X[i] * sqrt(2.0 * v[k]) + m[k]);
  }
  __builtin_amdgcn_wave_barrier();  // the table is written and read by this wave only (LDS ops are in order)
  int total = 1;
  for (int k = 0; k < D; ++k) total *= 10;
  const int label = (int)y;  // 1..K
  const bool valid = (y == (double)label) && label >= 1 && label <= K;
  double ve = 0.0;
  double hv[HMOGP_MAXJ], gx[HMOGP_MAXJ];
  const bool exact_dm = (quirks & HMOGP_QUIRK_CATEGORICAL_DM) == 0;
#pragma unroll
  for (int k = 0; k < HMOGP_MAXJ; ++k) hv[k] = gx[k] = 0.0;
  for (int n = lane; n < total; n += 64) {
    double e[HMOGP_MAXJ];
    double w = 1.0, esum = 0.0;
    int rem = n;
    // C-order grid (categorical.py:153-162): function 0 is the slowest index
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
    const double den = 1.0 + esum;
    // class probabilities, clipped then renormalised (:41-44)
    double psum = 0.0, py = 0.0;
#pragma unroll
    for (int k = 0; k < HMOGP_MAXJ; ++k) {
      if (k < D) {
        const double pk = clip(e[k] / den, 1e-9, 1.0 - 1e-9);
        psum += pk;
        if (label == k + 1) py = pk;
      }
    }
    const double pK = clip(1.0 / den, 1e-9, 1.0 - 1e-9);
    psum += pK;
    if (label == K) py = pK;
    ve += w * log(py / psum);
    // second derivative of log p wrt f_d (:115-128): -(e_d + sum_{j != d} e^{f_j + f_d}) / den^2, independent of y
    const double den2 = safe_square(den);
#pragma unroll
    for (int d = 0; d < HMOGP_MAXJ; ++d) {
      if (d < D) {
        double num = e[d];
#pragma unroll
        for (int j = 0; j < HMOGP_MAXJ; ++j)
          if (j < D && j != d) num += fmin(e[j] * e[d], 1.79769313486231570815e308);
        hv[d] += w * (num / den2);
        if (exact_dm) gx[d] += w * ((label == d + 1 ? 1.0 : 0.0) - e[d] / den);  // E[d log p_y / d f_d], softmax
      }
    }
  }
  o.ve = valid ? wave_sum(ve) : nan("");
  double wpow = 1.0;
  for (int k = 0; k < D; ++k) wpow *= GH10_WSUM_OVER_SQRTPI;
#pragma unroll
  for (int d = 0; d < HMOGP_MAXJ; ++d) {
    if (d < D) {
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
template <int LIK>
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
    lik_beta_wave(y, m, v, lane, o);
  else
    lik_categorical_wave(y, m, v, (int)param, lane, etab, quirks, o);
  if ((LIK == HMOGP_LIK_GAMMA || LIK == HMOGP_LIK_BETA) && !(quirks & HMOGP_QUIRK_GAMMA_BETA_PI)) {
    o.ve *= M_PI;  // exact mode: undo the second division of each dimension's weights by sqrt(pi) (quirk Q1)
#pragma unroll
    for (int j = 0; j < 2; ++j) o.gm[j] *= M_PI, o.gv[j] *= M_PI;
  }
}
