// rowpass.hip -- the row-streaming kernels of the svmogp_inf path (gfx950): everything that touches N x M data
// except the two FP64-MFMA contractions (gemm_f64.hip).
//
//   rbf_cross_cov   K^[n,m] = s2 exp(-r2/2)          HBM-bound writer (N*M*8 B), util.py:145-164 / GPy RBF.K
//   (row statistics p = K^ a, c = rowsum(P~ .* K^) ... are fused into the forward contraction's epilogue, gemm_f64.hip)
//   quad            q(f) mean/variance (svmogp_inf.py:212-218) -> variational expectations (het_likelihood.py:
//                   101-131) -> row weights alpha/beta for the backward pass + scalar statistics
//   colstats        r = K^T alpha, dZ numerators = colsum(E^ .* (x - z))                     (reads K^, P~ once)
//   reducers        deterministic two-level sums (block partials -> bundle), no floating-point atomics
#include <type_traits>

#include "rowpass.h"
#include "lik_device.h"
#include "rbf_device.h"

namespace {

constexpr int RBF_ROWS = 32;  // rows per block; 256 threads x 2 columns = 512 columns per block

// EXACT = GPy's rounding order r = sqrt(clip(r2))/l; K = s2 exp(-0.5 r*r)   (used for K_uu, whose inverse amplifies
// rounding);  !EXACT = s2 exp(-0.5 clip(r2) * (1/l^2)): no sqrt / divide per element -- the K_uf kernel is otherwise bound
// by the FP64 VALU (sqrt + divide cost as much as the exp), not by HBM.  The two differ by <= 2 ulp of the exponent.
template <int P, bool EXACT>
__global__ __launch_bounds__(256) void rbf_kernel(const double* __restrict__ X, int ldx, long long N,
                                                  const double* __restrict__ Z, int ldz, int M, double var, double ell,
                                                  double* __restrict__ K, int same, const int* __restrict__ rowwin,
                                                  RbfBatch bt) {
  if (bt.var) {  // batched over the latents (grid.z): per-latent inducing block, hyper-parameters, output and windows
    const int q = blockIdx.z;
    X += (long long)q * bt.sX;
    Z += (long long)q * bt.sZ;
    K += (long long)q * bt.sK;
    var = bt.var[q], ell = bt.ell[q];
    if (rowwin) rowwin += (long long)q * bt.sWin;
  }
  __shared__ double xs[RBF_ROWS][P + 1];
  const int t = threadIdx.x;
  const long long n0 = (long long)blockIdx.x * RBF_ROWS;
  const int c = blockIdx.y * 512 + 2 * t;
  for (int e = t; e < RBF_ROWS * P; e += 256) {
    const int r = e / P, p = e % P;
    xs[r][p] = (n0 + r < N) ? X[(n0 + r) * ldx + p] : 0.0;
  }
  __syncthreads();
  if (t < RBF_ROWS) {
    double xv[P];
#pragma unroll
    for (int p = 0; p < P; ++p) xv[p] = xs[t][p];
    xs[t][P] = sumsq<P>(xv);
  }
  __syncthreads();
  if (c >= M) return;
  if (rowwin) {  // exact-zero windows: a wave owns one 128-column block; skip it if no consumer will read it
    const int T = (int)(n0 >> 7), cb0 = (c >> 7) << 7;
    if (cb0 >= rowwin[2 * T + 1] || cb0 + 128 <= rowwin[2 * T]) return;
  }
  const bool two = (c + 1) < M;
  double z0[P], z1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    z0[p] = Z[(long long)c * ldz + p];
    z1[p] = two ? Z[(long long)(c + 1) * ldz + p] : 0.0;
  }
  const double zs0 = sumsq<P>(z0), zs1 = sumsq<P>(z1);
  const bool vec = two && ((M & 1) == 0);
  const int nr = (int)min((long long)RBF_ROWS, N - n0);
  // (the row loop is instantiated for the vector-store and the scalar-store case: no per-row branch on a loop-invariant condition)
  auto rows = [&](auto vec_c) {
    constexpr bool VEC = decltype(vec_c)::value;
    for (int r = 0; r < nr; ++r) {
      double xv[P];
#pragma unroll
      for (int p = 0; p < P; ++p) xv[p] = xs[r][p];
      const double xsq = xs[r][P];
      double r20, r21;
      if (EXACT) {
        r20 = rbf_r2<P>(xv, xsq, z0, zs0, ell), r21 = rbf_r2<P>(xv, xsq, z1, zs1, ell);
      } else {
        const double il2 = 1.0 / (ell * ell);
        r20 = rbf_r2_fast<P>(xv, xsq, z0, zs0, il2), r21 = rbf_r2_fast<P>(xv, xsq, z1, zs1, il2);
      }
      if (EXACT && same) {  // GPy's X2=None branch forces the diagonal distance to 0 (kern/stationary; K_uu-side calls only)
        if (n0 + r == c) r20 = 0.0;
        if (n0 + r == c + 1) r21 = 0.0;
      }
      // exp(-r2/2) underflows to exactly 0.0 beyond r2 ~ 1490.3: a wave whose 128 columns are all past that writes the
      // zeros without evaluating the exponential (same values; NaNs fail the comparison and take the full path)
      double k0 = 0.0, k1 = 0.0;
      if (!__all(r20 > 1492.0 && r21 > 1492.0)) k0 = var * exp(-0.5 * r20), k1 = var * exp(-0.5 * r21);
      double* out = K + (n0 + r) * M + c;
      if (VEC)
        *reinterpret_cast<f64x2*>(out) = f64x2{k0, k1};
      else {
        out[0] = k0;
        if (two) out[1] = k1;
      }
    }
  };
  if (vec) rows(std::true_type{});
  else rows(std::false_type{});
}

// ---- exact-zero windows ----------------------------------------------------------------------------------------
// K^[n][m] = s2 exp(-r2/2) is EXACTLY 0.0 in float64 once r2 > ~1490.3 (exp underflows below the smallest subnormal).
// For spatially sorted inputs K^ is therefore banded, and every product the row pass forms with an entry outside the
// band is a product with an exact zero.  window_kernel finds, per 128-row tile, the column range [lo, hi) that holds all
// possibly-nonzero entries (distance to the tile's bounding box <= sqrt(1491) lengthscales: a superset), and per
// 128-column block the hull of the row tiles that touch it;
// window_fix_kernel falls back to full ranges when the data is not banded (a row tile inside a block's hull that does not
// touch the block), so arbitrary inputs stay correct.  No host synchronisation is involved.
constexpr double WINDOW_R2_MAX = 1491.0;

__global__ void window_init_kernel(int* __restrict__ colwin, int ncb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncb) {
    colwin[2 * i] = 0x7fffffff;
    colwin[2 * i + 1] = 0;
  }
}

template <int P>
__global__ __launch_bounds__(256) void window_kernel(const double* __restrict__ X, long long N, const double* __restrict__ Z,
                                                     int ldz, int M, double ell, int* __restrict__ rowwin,
                                                     int* __restrict__ colwin, unsigned char* __restrict__ hit, int ncb) {
  // A column m can hold a non-zero for some row of this 128-row tile only if z_m is within sqrt(1491) lengthscales of the
  // tile's bounding box (per-dimension [min, max] of its inputs): a conservative test costing O(M) per tile.  Sorted
  // 1-D inputs give tight boxes (narrow windows); unsorted inputs give boxes spanning the domain (full windows).
  __shared__ double bmin[P], bmax[P];
  __shared__ double red[2][4][P];
  __shared__ int s_lo, s_hi;
  __shared__ int s_hit[64];
  const int t = threadIdx.x, T = blockIdx.x, lane = t & 63, w = t >> 6;
  const long long r0 = (long long)T * 128;
  const int nr = (int)min(128LL, N - r0);
  if (t == 0) s_lo = 0x7fffffff, s_hi = 0;
  if (t < 64) s_hit[t] = 0;
  double lo[P], hi[P];
#pragma unroll
  for (int p = 0; p < P; ++p) lo[p] = INFINITY, hi[p] = -INFINITY;
  bool nan_in = false;
  if (t < nr) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double x = X[(r0 + t) * P + p];
      lo[p] = hi[p] = x;
      nan_in |= (x != x);
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[p] = fmin(lo[p], __shfl_xor(lo[p], o, 64));
      hi[p] = fmax(hi[p], __shfl_xor(hi[p], o, 64));
    }
    if (lane == 0) red[0][w][p] = lo[p], red[1][w][p] = hi[p];
  }
  const bool any_nan = __syncthreads_or(nan_in ? 1 : 0) != 0;
  if (t == 0) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bmin[p] = fmin(fmin(red[0][0][p], red[0][1][p]), fmin(red[0][2][p], red[0][3][p]));
      bmax[p] = fmax(fmax(red[1][0][p], red[1][1][p]), fmax(red[1][2][p], red[1][3][p]));
    }
  }
  __syncthreads();
  const double thr = WINDOW_R2_MAX * ell * ell * (1.0 + 1e-9);
  for (int m = t; m < M; m += 256) {
    double d2 = 0.0;
    bool zn = false;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double z = Z[(long long)m * ldz + p];
      zn |= (z != z);
      const double d = fmax(fmax(bmin[p] - z, z - bmax[p]), 0.0);  // distance from z to the box along dimension p
      d2 += d * d;
    }
    if (any_nan || zn || !(d2 > thr)) {  // NaNs count as "possibly non-zero"
      atomicMin(&s_lo, m);
      atomicMax(&s_hi, m + 1);
      s_hit[m >> 7] = 1;
    }
  }
  __syncthreads();
  if (t == 0) {
    const bool none = s_hi == 0;
    rowwin[2 * T] = none ? 0 : (s_lo & ~15);
    rowwin[2 * T + 1] = none ? 0 : min((s_hi + 15) & ~15, (M + 15) & ~15);
  }
  if (t < ncb) {
    hit[(long long)T * ncb + t] = (unsigned char)s_hit[t];
    if (s_hit[t]) {
      atomicMin(&colwin[2 * t], (int)r0);
      atomicMax(&colwin[2 * t + 1], (int)(r0 + nr));
    }
  }
}

__global__ __launch_bounds__(256) void window_fix_kernel(int* __restrict__ rowwin, int* __restrict__ colwin,
                                                         const unsigned char* __restrict__ hit, int tiles, int ncb, int M,
                                                         long long N) {
  __shared__ int viol;
  const int t = threadIdx.x;
  if (t == 0) viol = 0;
  __syncthreads();
  for (int cb = 0; cb < ncb; ++cb) {
    const int lo = colwin[2 * cb], hi = colwin[2 * cb + 1];
    if (lo == 0x7fffffff) continue;
    for (int T = (lo >> 7) + t; T < (hi + 127) >> 7; T += 256)
      if (!hit[(long long)T * ncb + cb]) viol = 1;
  }
  __syncthreads();
  if (viol) {  // not banded: dense ranges everywhere
    for (int T = t; T < tiles; T += 256) {
      rowwin[2 * T] = 0;
      rowwin[2 * T + 1] = (M + 15) & ~15;
    }
    for (int cb = t; cb < ncb; cb += 256) {
      colwin[2 * cb] = 0;
      colwin[2 * cb + 1] = (int)N;
    }
  } else {
    for (int cb = t; cb < ncb; cb += 256)
      if (colwin[2 * cb] == 0x7fffffff) colwin[2 * cb] = 0, colwin[2 * cb + 1] = 0;
  }
}

// ---- quad -----------------------------------------------------------------------------------------------
// LDS of one quadrature block
struct QuadShared {
  double etab[4][HMOGP_ETAB];
  double red[4][HMOGP_MAXSCAL];
  // mixing weights of this task's functions: from the kernel arguments, or (captured-graph replays) from device memory
  double w[HMOGP_MAXQ][HMOGP_MAXJ], w0[HMOGP_MAXQ][HMOGP_MAXJ], kap[HMOGP_MAXQ][HMOGP_MAXJ], var[HMOGP_MAXQ];
  double scale;
};

// The quadrature of one block of rows (block `blk` of the segment `a` describes).  DEVONLY: the by-value weights of `a` do not
// exist (quad_multi_kernel).
template <int LIK, int CATD, bool DEVONLY>
__device__ __forceinline__ void quad_body(const QuadArgs& a, unsigned blk, QuadShared& sh) {
  constexpr int G = lik_lanes(LIK);
  auto& etab = sh.etab;
  auto& red = sh.red;
  auto& s_w = sh.w;
  auto& s_w0 = sh.w0;
  auto& s_kap = sh.kap;
  auto& s_var = sh.var;
  double& s_scale = sh.scale;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long long n = ((long long)blk * 256 + t) / G;
  const bool valid = n < a.N;                       // uniform per wave when G == 64
  // (row of the pool-wide [Q][ldn] vectors: `a.off` is the segment's offset in the pool when the pointers are NOT pre-offset --
  //  quad_multi_kernel: eight derived pointers per segment would be 16 more live SGPRs, and the instantiations that carry four or
  //  five likelihood bodies spilled SGPRs to scratch memory, 68 bytes per lane)
  const long long nn = n + a.off;
  const bool lead = valid && (G == 1 || lane == 0);  // the lane that owns the row's outputs
  const int Q = a.Q, J = a.dimf;
  const int nscal = 2 + 2 * Q + J + Q * J;
  // the row's inputs are requested BEFORE the weights are staged: one memory round trip for both (small models: the kernel is
  // a chain of latencies)
  double mu[HMOGP_MAXJ], vv[HMOGP_MAXJ], pq[HMOGP_MAXQ], cq[HMOGP_MAXQ];
#pragma unroll
  for (int q = 0; q < HMOGP_MAXQ; ++q) {
    pq[q] = (valid && q < Q) ? a.p[q * a.ldn + nn] : 0.0;
    cq[q] = (valid && q < Q) ? a.c[q * a.ldn + nn] : 0.0;
  }
  const double yv = valid ? a.y[n] : 0.0, yauxv = (valid && a.yaux) ? a.yaux[n] : 0.0;
  if (t < HMOGP_MAXQ * HMOGP_MAXJ) {
    const int q = t / HMOGP_MAXJ, j = t % HMOGP_MAXJ;
    const bool in = q < a.Q && j < a.dimf;
    if (DEVONLY || a.Wd) {
      const long long o = (long long)q * a.Df + a.d0 + j;
      s_w[q][j] = in ? a.Wd[o] : 0.0, s_w0[q][j] = in ? a.W0d[o] : 0.0, s_kap[q][j] = in ? a.kapd[o] : 0.0;
      if (j == 0) s_var[q] = q < a.Q ? a.vard[q] : 0.0;
      if (t == 0) s_scale = a.scaled[0];
    } else if (!DEVONLY) {
      s_w[q][j] = a.w[q][j], s_w0[q][j] = a.w0[q][j], s_kap[q][j] = a.kap[q][j];
      if (j == 0) s_var[q] = a.var[q];
      if (t == 0) s_scale = a.scale;
    }
  }
  __syncthreads();
  // contribution of this lane to block scalar `slot` (uniform slot; every lane of the wave calls)
  auto emit = [&](int slot, double val) {
    const double s = (G == 1) ? wave_sum(val) : val;
    if (lane == 0) red[w][slot] = s;
  };
  bool neg = false;
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) {
    double m = 0.0, v = 0.0;
    if (j < J) {
#pragma unroll
      for (int q = 0; q < HMOGP_MAXQ; ++q)
        if (q < Q) {
          const double wq = s_w[q][j];
          m += wq * pq[q];
          v += (wq * wq + s_kap[q][j]) * s_var[q] + wq * wq * cq[q];
        }
      neg |= (v < 0.0);
    }
    mu[j] = m;
    vv[j] = v;
  }
  LikOut o;
  o.ve = 0.0;
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) o.gm[j] = o.gv[j] = 0.0;
  if (valid) lik_eval<LIK, CATD>(yv, yauxv, mu, vv, a.lik_param, lane, etab[w], a.quirks, o);
  if (G == 64 || a.pg) {  // wave-per-row likelihoods: p / c were only needed for q(f); re-read them instead of keeping 2 x MAXQ
                          // doubles alive across the node loop (register pressure = occupancy of the quadrature).
                          // Strict q(f) (a.pg != nullptr): the block scalars sa / swk take the explicit-inverse forms K^ a and
                          // rowsum(P~ .* K^) the reference's gradient code uses (svmogp_inf.py:157-161), q(f) took the solve-based ones
    const double* pp = a.pg ? a.pg : a.p;
    const double* cc = a.pg ? a.cg : a.c;
#pragma unroll
    for (int q = 0; q < HMOGP_MAXQ; ++q) {
      pq[q] = (valid && q < Q) ? pp[q * a.ldn + nn] : 0.0;
      cq[q] = (valid && q < Q) ? cc[q * a.ldn + nn] : 0.0;
    }
  }
  // (the loops over a row's functions are unrolled to HMOGP_MAXJ with a guard: a run-time trip count indexes mu / vv / the LikOut
  //  arrays dynamically and parks them in scratch -- 144 bytes per lane in EVERY quad_kernel / var_exp_kernel instantiation, r5)
  if (a.out_mu && lead) {
#pragma unroll
    for (int j = 0; j < HMOGP_MAXJ; ++j)
      if (j < J) {
        a.out_mu[n * J + j] = mu[j];
        a.out_v[n * J + j] = vv[j];
      }
  }
  const double s = lead ? s_scale : 0.0;  // non-owning lanes contribute zeros
  o.ve *= s;
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) {
    o.gm[j] *= s;
    o.gv[j] *= s;
  }
  if (a.out_gm && lead) {
#pragma unroll
    for (int j = 0; j < HMOGP_MAXJ; ++j)
      if (j < J) {
        a.out_gm[n * J + j] = o.gm[j];
        a.out_gv[n * J + j] = o.gv[j];
      }
  }
  emit(0, o.ve);
  emit(1, (lead && neg) ? 1.0 : 0.0);
#pragma unroll
  for (int q = 0; q < HMOGP_MAXQ; ++q) {
    if (q < Q) {
      double al = 0.0, be = 0.0, al0 = 0.0, be0 = 0.0;
#pragma unroll
      for (int j = 0; j < HMOGP_MAXJ; ++j)
        if (j < J) {
          const double wq = s_w[q][j], w0 = s_w0[q][j];
          al += wq * o.gm[j];
          be += wq * wq * o.gv[j];
          al0 += w0 * o.gm[j];
          be0 += w0 * wq * o.gv[j];
          emit(2 + 2 * Q + J + q * J + j, o.gm[j] * pq[q] + 2.0 * wq * (o.gv[j] * cq[q]));  // swk[q][j]
        }
      if (lead) {
        a.alpha[q * a.ldn + nn] = al;
        a.beta[q * a.ldn + nn] = be;
        a.alpha0[q * a.ldn + nn] = al0;
        a.beta0[q * a.ldn + nn] = be0;
      }
      const double ptq = (lead && a.pt) ? a.pt[q * a.ldn + nn] : 0.0;
      const double ctq = (lead && a.ct) ? a.ct[q * a.ldn + nn] : 0.0;
      emit(2 + 2 * q, al0 * pq[q] + 2.0 * be0 * cq[q]);  // sa_q
      emit(3 + 2 * q, al0 * ptq + 2.0 * be0 * ctq);      // sl_q
    }
  }
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j)
    if (j < J) emit(2 + 2 * Q + j, o.gv[j]);  // sgv[j]
  __syncthreads();
  if (t < nscal) a.partials[(long long)blk * nscal + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

template <int LIK, int CATD = 0>
__global__ __launch_bounds__(256) void quad_kernel(QuadArgs a) {
  __shared__ QuadShared sh;
  quad_body<LIK, CATD, false>(a, blockIdx.x, sh);
}

// All segments of a pool in ONE launch (small models: three likelihood families of ~1000 rows each are three launches of a few
// blocks otherwise); block -> segment by the prefix of block counts, likelihood by a block-uniform branch.
// [r5] MASK = the likelihood families this instantiation carries (bit L for family L, bit 8 + d for Categorical with d latent
// functions).  One kernel with all fifteen bodies inlined is allocated for the worst of them (256 VGPRs + 32 AGPRs, 380 spilled
// SGPRs, one wave per SIMD); the sets of the BASELINE configurations get instantiations of their own (C1's
// {HetGaussian, Bernoulli, Categorical(3)}: see the register table in DESIGN 11e), any other set the all-inclusive one.
constexpr unsigned qm_bit(int lik, int dimf) { return lik == HMOGP_LIK_CATEGORICAL ? 1u << (8 + dimf) : 1u << lik; }
constexpr unsigned QM_C1 = qm_bit(HMOGP_LIK_HETGAUSSIAN, 0) | qm_bit(HMOGP_LIK_BERNOULLI, 0) | qm_bit(HMOGP_LIK_CATEGORICAL, 2);
constexpr unsigned QM_H4 = qm_bit(HMOGP_LIK_GAUSSIAN, 0) | qm_bit(HMOGP_LIK_BERNOULLI, 0) | qm_bit(HMOGP_LIK_POISSON, 0) |
                           qm_bit(HMOGP_LIK_GAMMA, 0);
constexpr unsigned QM_C5 = qm_bit(HMOGP_LIK_GAUSSIAN, 0) | qm_bit(HMOGP_LIK_CATEGORICAL, 3);
constexpr unsigned QM_LIGHT = qm_bit(HMOGP_LIK_GAUSSIAN, 0) | qm_bit(HMOGP_LIK_BERNOULLI, 0) | qm_bit(HMOGP_LIK_HETGAUSSIAN, 0) |
                              qm_bit(HMOGP_LIK_POISSON, 0) | qm_bit(HMOGP_LIK_EXPONENTIAL, 0);

template <unsigned MASK>
__global__ __launch_bounds__(256) void quad_multi_kernel(QuadMulti m) {
  __shared__ QuadShared sh;
  int s = 0;
  while (s + 1 < m.nseg && blockIdx.x >= m.seg[s + 1].blk0) ++s;
  const QuadSeg& g = m.seg[s];
  QuadArgs a;
  a.lik = g.lik, a.lik_param = g.lik_param, a.dimf = g.dimf, a.Q = m.Q, a.N = g.N;
  a.y = g.y, a.yaux = g.yaux;
  a.p = m.p, a.c = m.c, a.pt = m.pt, a.ct = m.ct, a.off = g.off;
  a.ldn = m.ldn;
  a.Wd = m.Wd, a.W0d = m.W0d, a.kapd = m.kapd, a.vard = m.vard, a.scaled = m.scale_base + g.t;
  a.Df = m.Df, a.d0 = g.d0;
  a.pg = a.cg = nullptr;
  a.quirks = m.quirks;
  a.alpha = m.alpha, a.beta = m.beta, a.alpha0 = m.alpha0, a.beta0 = m.beta0;
  a.partials = m.partials + g.part0;
  a.out_mu = a.out_v = a.out_gm = a.out_gv = nullptr;
  const unsigned blk = blockIdx.x - g.blk0;
#define QB(L)                                                  \
  if constexpr ((MASK & qm_bit(L, 0)) != 0) {                  \
    if (g.lik == L) {                                          \
      quad_body<L, 0, true>(a, blk, sh);                       \
      return;                                                  \
    }                                                          \
  }
#define QBC(D)                                                 \
  if constexpr ((MASK & qm_bit(HMOGP_LIK_CATEGORICAL, D)) != 0) { \
    if (g.lik == HMOGP_LIK_CATEGORICAL && g.dimf == D) {       \
      quad_body<HMOGP_LIK_CATEGORICAL, D, true>(a, blk, sh);   \
      return;                                                  \
    }                                                          \
  }
  QB(HMOGP_LIK_GAUSSIAN) QB(HMOGP_LIK_BERNOULLI) QB(HMOGP_LIK_HETGAUSSIAN) QB(HMOGP_LIK_POISSON) QB(HMOGP_LIK_EXPONENTIAL)
  QB(HMOGP_LIK_GAMMA) QB(HMOGP_LIK_BETA)
  QBC(1) QBC(2) QBC(3) QBC(4) QBC(5) QBC(6) QBC(7) QBC(8)
#undef QB
#undef QBC
}


// ---- colstats: thread = 2 columns, block = 512 columns x `rows` rows ----------------------------------------
// STRICT (HMOGP_CFG_STRICT_QF only): r = A^T alpha from a second matrix and quirk Q10's r == 0 gating -- kept out of the hot-path
// instantiation, which has to fit 64 registers to run beside the Gram.
// [r6] WIN = false (no exact-zero windows: the default): the row loop's bounds are the same for every lane, so what a row contributes
// besides its K^ / P~ entries -- alpha, alpha0, beta0 and the input x_n -- is WAVE-UNIFORM.  Read through the constant address
// space at uniform addresses those values arrive in SGPRs (s_load; v_fma_f64 takes an SGPR pair as a source) instead of being
// broadcast into 6 + 2 P vector registers per lane: P = 2 ... 4 fit their 64 registers without a spill (2 - 64 spilled before; P = 2
// is BASELINE config 5's path: column statistics 7.6 -> 4.6 ms per step there).  WIN = true keeps the per-lane bounds and vector
// loads: the windows mode, and P = 1, which never spilled and is 10 % faster with them (launch_colstats).
#define HM_CONST(p) ((const __attribute__((address_space(4))) double*)(uintptr_t)(p))
template <int P, bool STRICT, bool SL, bool WIN>
__global__ __launch_bounds__(256, 8) void colstats_kernel(const double* __restrict__ Kh, const double* __restrict__ Pt,
                                                       const double* __restrict__ a, const double* __restrict__ alpha,
                                                       const double* __restrict__ alpha0, const double* __restrict__ beta0,
                                                       const double* __restrict__ X, const double* __restrict__ Z, int ldz,
                                                       long long N, int M, int rows, int want_z,
                                                       double* __restrict__ partials, const int* __restrict__ colwin,
                                                       ColBatch bt, const double* __restrict__ Ar, const double* __restrict__ ell) {
  {  // batched over the latents (grid.z)
    const long long q = blockIdx.z;
    if (STRICT) Ar += q * bt.sK;
    Kh += q * bt.sK, Pt += q * bt.sK, a += q * bt.sA, alpha += q * bt.sV, alpha0 += q * bt.sV, beta0 += q * bt.sV;
    Z += q * bt.sZ, partials += q * bt.sPart;
    if (colwin) colwin += q * bt.sWin;
  }
  const int t = threadIdx.x, c = blockIdx.x * 512 + 2 * t;
  if (c >= M) return;
  const bool two = (c + 1) < M;
  const bool vec = two && ((M & 1) == 0);
  double z0[P], z1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    z0[p] = Z[(long long)c * ldz + p];
    z1[p] = two ? Z[(long long)(c + 1) * ldz + p] : 0.0;
  }
  const double a0 = a[c], a1 = two ? a[c + 1] : 0.0;
  const double zs0 = STRICT ? sumsq<P>(z0) : 0.0, zs1 = STRICT ? sumsq<P>(z1) : 0.0;
  // [r5] ell != nullptr: also  s2[m] = sum_n E_nm |x_n - z_m|^2 / l^2  -- the r2-weighted statistic behind the lengthscale gradient
  // (sl = sum_m s2[m]; svmogp.py:140 -> GPy update_gradients_full).  It used to be carried by the forward contraction's epilogue
  // as two more row statistics (p~, c~) with the distances recomputed per element there; E_nm and x_n - z_m are in hand here.
  const bool want_e = want_z || SL;
  // A block takes the row splits blockIdx.y, blockIdx.y + gridDim.y, ...: the grid may be CAPPED (launch_colstats) so that
  // this HBM-bound pass occupies only a few CU slots at a time beside the FP64-MFMA Gram it runs next to -- a block that
  // holds half a CU while it waits for HBM keeps a Gram block (whose registers fill the other half) from being scheduled.
  const long long nsplit = (N + rows - 1) / rows;
  for (long long sp = blockIdx.y; sp < nsplit; sp += gridDim.y) {
    long long n0 = sp * rows, n1 = min(N, n0 + rows);
    if (WIN && colwin) {  // rows outside the exact-zero window of this 128-column block contribute exact zeros
      n0 = max(n0, (long long)colwin[2 * (c >> 7)]);
      n1 = min(n1, (long long)colwin[2 * (c >> 7) + 1]);
    }
    double r0 = 0.0, r1 = 0.0, d0[P], d1[P], s20 = 0.0, s21 = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) d0[p] = d1[p] = 0.0;
    for (long long n = n0; n < n1; ++n) {
      double k0, k1, q0 = 0.0, q1 = 0.0;
      if (vec) {
        const f64x2 kv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(Kh + n * M + c));
        k0 = kv.x, k1 = kv.y;
        if (want_e) {
          const f64x2 qv = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(Pt + n * M + c));
          q0 = qv.x, q1 = qv.y;
        }
      } else {
        k0 = Kh[n * M + c];
        k1 = two ? Kh[n * M + c + 1] : 0.0;
        if (want_e) {
          q0 = Pt[n * M + c];
          q1 = two ? Pt[n * M + c + 1] : 0.0;
        }
      }
      const double al = WIN ? alpha[n] : HM_CONST(alpha)[n];
      if (STRICT) {   // strict q(f): r = A^T alpha with A = K^ Kuu^-1 (dVE_dmu of svmogp_inf.py:144 as the reference forms it)
        r0 += Ar[n * M + c] * al;
        r1 += (two ? Ar[n * M + c + 1] : 0.0) * al;
      } else {
        r0 += k0 * al;
        r1 += k1 * al;
      }
      if (want_e) {
        const double al0 = WIN ? alpha0[n] : HM_CONST(alpha0)[n], be0 = 2.0 * (WIN ? beta0[n] : HM_CONST(beta0)[n]);
        double e0 = (al0 * a0 + be0 * q0) * k0, e1 = (al0 * a1 + be0 * q1) * k1;
        // Quirk Q10 (STRICT only): GPy's gradients_X drops the entries whose COMPUTED distance -- the expanded form |x|^2 + |z|^2 -
        // 2 x.z, clipped -- is exactly 0.  That needs |x - z|^2 below a few ulp of |z|^2: un-centred inputs, or, at 1e6 rows per
        // task, the odd row within 1.5e-8 |z| of an inducing point (one such row in the full-size C4 test: 4e-8 of one g_Z entry).
        // The default path keeps those terms (they are the mathematically correct ones): an in-loop test, even behind a cheap
        // pre-test, costs this kernel 16 spilled registers -- and it has to fit 64 to run beside the Gram.
        double g20 = 0.0, g21 = 0.0;     // STRICT: GPy's clipped expanded-form squared distances (unscaled)
        if (STRICT) {
          double xv[P];
#pragma unroll
          for (int p = 0; p < P; ++p) xv[p] = WIN ? X[n * P + p] : HM_CONST(X)[n * P + p];
          const double xsq = sumsq<P>(xv);
          g20 = rbf_r2_fast<P>(xv, xsq, z0, zs0, 1.0), g21 = rbf_r2_fast<P>(xv, xsq, z1, zs1, 1.0);
          if (g20 == 0.0) e0 = 0.0;
          if (g21 == 0.0) e1 = 0.0;
        }
        double q20 = 0.0, q21 = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const double x = WIN ? X[n * P + p] : HM_CONST(X)[n * P + p];
          const double dx0 = x - z0[p], dx1 = x - z1[p];
          d0[p] += e0 * dx0;
          d1[p] += e1 * dx1;
          if (SL) q20 += dx0 * dx0, q21 += dx1 * dx1;
        }
        // [r6] strict q(f): the r2 weight of the lengthscale statistic in GPy's OWN form -- |x|^2 + |z|^2 - 2 x.z, clipped
        // (stationary.py `_unscaled_dist`): with un-centred inputs that form loses digits the reference's gradient carries
        // (lad_c1_offset_rung1) -- instead of sum_p (x_p - z_p)^2.  Replaces the r2-weighted twins of strict_rowstats_kernel.
        if (SL) s20 += e0 * (STRICT ? g20 : q20), s21 += e1 * (STRICT ? g21 : q21);
      }
    }
    // partial layout per row-split: [ r (M) | dZ (M*P) | s2 (M) ]
    double* out = partials + sp * ((long long)M * (2 + P));
    const double il2 = ell ? 1.0 / (ell[blockIdx.z] * ell[blockIdx.z]) : 0.0;   // (scaled once per split, not per element)
    out[(long long)M * (1 + P) + c] = s20 * il2;
    if (two) out[(long long)M * (1 + P) + c + 1] = s21 * il2;
    out[c] = r0;
    if (two) out[c + 1] = r1;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      out[M + (long long)c * P + p] = d0[p];
      if (two) out[M + (long long)(c + 1) * P + p] = d1[p];
    }
  }
}

// ---- reducers ---------------------------------------------------------------------------------------------
// dst[off[k]] += sum_b partials[b*len + k]   (off == nullptr: dst[k])
__global__ __launch_bounds__(256) void reduce_rows_kernel(const double* __restrict__ partials, long long nrows, int len,
                                                          const long long* __restrict__ off, double* __restrict__ dst,
                                                          int accumulate) {
  __shared__ double scratch[16];
  const int k = blockIdx.x;
  double s = 0.0;
  for (long long b = threadIdx.x; b < nrows; b += blockDim.x) s += partials[b * len + k];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) {
    double* d = dst + (off ? off[k] : k);
    *d = accumulate ? (*d + s) : s;
  }
}
// dst[i] (+)= sum_s slabs[s*stride + i], i < len.  Block = 16 columns x 16 slab lanes: thread (c, g) sums the slabs
// g, g+16, ... of column c (independent loads, 128-byte segments), the 16 partial sums are added in a fixed order.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const double* __restrict__ slabs, int nslabs, long long stride,
                                                           long long len, double* __restrict__ dst, int accumulate,
                                                           long long sSlabs, long long sDst) {
  __shared__ double part[16][17];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + c;
  slabs += (long long)blockIdx.y * sSlabs;
  dst += (long long)blockIdx.y * sDst;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < len) {
    int b = g;
    for (; b + 48 < nslabs; b += 64) {
      s0 += slabs[b * stride + i];
      s1 += slabs[(b + 16) * stride + i];
      s2 += slabs[(b + 32) * stride + i];
      s3 += slabs[(b + 48) * stride + i];
    }
    for (; b < nslabs; b += 16) s0 += slabs[b * stride + i];
  }
  part[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < len) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += part[k][c];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}
// dst[i][j] (+)= sum_s slabs[s][i][j] over the LOWER 128 x 128 tiles only (the Gram product never writes the others)
__global__ __launch_bounds__(256) void reduce_slabs_lower_kernel(const double* __restrict__ slabs, int nslabs, int M,
                                                                 double* __restrict__ dst, int accumulate, long long sSlabs,
                                                                 long long sDst) {
  slabs += (long long)blockIdx.z * sSlabs;
  dst += (long long)blockIdx.z * sDst;
  int v = blockIdx.x, ti = (int)((sqrt(8.0 * (double)v + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= v) ++ti;
  while (ti * (ti + 1) / 2 > v) --ti;
  const int tj = v - ti * (ti + 1) / 2;
  const long long MM = (long long)M * M;
  const int c2 = (threadIdx.x & 63) * 2, r4 = threadIdx.x >> 6;  // 64 lanes x 2 columns; block = 4 rows (blockIdx.y)
  const int col = tj * 128 + c2;
  if (col >= M) return;
  const bool two = col + 1 < M, vec = two && ((M & 1) == 0);
  {
    const int row = ti * 128 + blockIdx.y * 4 + r4;
    if (row >= M) return;
    const long long off = (long long)row * M + col;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nslabs; ++b) {
      if (vec) {
        const f64x2 x = *reinterpret_cast<const f64x2*>(slabs + b * MM + off);
        s0 += x.x, s1 += x.y;
      } else {
        s0 += slabs[b * MM + off];
        if (two) s1 += slabs[b * MM + off + 1];
      }
    }
    dst[off] = accumulate ? dst[off] + s0 : s0;
    if (two) dst[off + 1] = accumulate ? dst[off + 1] + s1 : s1;
  }
}
// lower tiles of H were computed: mirror to the upper triangle
__global__ void mirror_lower_kernel(double* __restrict__ A, int M, long long stride) {
  double* a = A + (long long)blockIdx.z * stride;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < M && c > r) a[(long long)r * M + c] = a[(long long)c * M + r];
}

// Wire format of the statistic bundle (what a multi-GPU run all-reduces): H_q is symmetric and the row pass only fills its
// lower triangle, so only that triangle travels -- [head NG | per q: tril(H_q) row-major packed (M(M+1)/2) | tail (per_q - M*M)].
// dir 0: bundle -> wire, dir 1: wire -> bundle (lower triangle; hmogp_step_finish mirrors it).
__global__ void wire_tri_kernel(double* __restrict__ bundle_q0, double* __restrict__ wire_q0, int M, long long per_q,
                                long long wire_per_q, int dir) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c > r) return;
  double* h = bundle_q0 + (long long)blockIdx.z * per_q + (long long)r * M + c;
  double* w = wire_q0 + (long long)blockIdx.z * wire_per_q + (long long)r * (r + 1) / 2 + c;
  if (dir == 0) *w = *h;
  else *h = *w;
}
// head (q == gridDim.y - 1 ... see launcher) and per-latent tails
__global__ void wire_rest_kernel(double* __restrict__ bundle, double* __restrict__ wire, long long NG, long long MM,
                                 long long Mtri, long long per_q, long long wire_per_q, int Q, int dir) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tail = per_q - MM;
  if (i < NG) {
    if (dir == 0) wire[i] = bundle[i];
    else bundle[i] = wire[i];
    return;
  }
  const long long k = i - NG;
  if (k >= tail * Q) return;
  const long long q = k / tail, e = k - q * tail;
  double* b = bundle + NG + q * per_q + MM + e;
  double* w = wire + NG + q * wire_per_q + Mtri + e;
  if (dir == 0) *w = *b;
  else *b = *w;
}

__global__ void raw_kmn_kernel(const double* __restrict__ a, const double* __restrict__ gm, const double* __restrict__ gv,
                               int J, int j, double w, const double* __restrict__ Pt, int M, long long N,
                               double* __restrict__ out) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (n >= N) return;
  out[(long long)m * N + n] = a[m] * gm[n * J + j] + 2.0 * w * (gv[n * J + j] * Pt[n * M + m]);
}

__global__ void gammaln1p_kernel(const double* __restrict__ y, double* __restrict__ out, long long N) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < N) out[i] = lgamma(y[i] + 1.0);
}

// ---- stand-alone variational expectations (hmogp_var_exp; also the predictive building block) -------------------
template <int LIK, int CATD = 0>
__global__ __launch_bounds__(256) void var_exp_kernel(int J, double param, long long N, const double* __restrict__ y,
                                                      const double* __restrict__ m, const double* __restrict__ v,
                                                      double* __restrict__ ve, double* __restrict__ dm,
                                                      double* __restrict__ dv, unsigned quirks) {
  constexpr int G = lik_lanes(LIK);
  __shared__ double etab[4][HMOGP_ETAB];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long long n = ((long long)blockIdx.x * 256 + t) / G;
  if (n >= N) return;
  double mu[HMOGP_MAXJ], vv[HMOGP_MAXJ];
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) {
    mu[j] = (j < J) ? m[n * J + j] : 0.0;
    vv[j] = (j < J) ? v[n * J + j] : 0.0;
  }
  LikOut o;
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) o.gm[j] = o.gv[j] = 0.0;
  const double yy = y[n];
  lik_eval<LIK, CATD>(yy, (LIK == HMOGP_LIK_POISSON) ? lgamma(yy + 1.0) : 0.0, mu, vv, param, lane, etab[w], quirks, o);
  if (G == 1 || lane == 0) {
    ve[n] = o.ve;
#pragma unroll
    for (int j = 0; j < HMOGP_MAXJ; ++j)
      if (j < J) {
        dm[n * J + j] = o.gm[j];
        dv[n * J + j] = o.gv[j];
      }
  }
}

// ---- predictive moments of y (hmogp_predictive) ------------------------------------------------------------------
template <int LIK>
__global__ __launch_bounds__(256) void predictive_kernel(int J, int Jp, double param, int T, long long N,
                                                         const double* __restrict__ m, const double* __restrict__ v,
                                                         double* __restrict__ mean, double* __restrict__ var) {
  constexpr int G = lik_pred_lanes(LIK);
  __shared__ double etab[4][HMOGP_ETAB];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long long n = ((long long)blockIdx.x * 256 + t) / G;
  if (n >= N) return;
  double mu[HMOGP_MAXJ], vv[HMOGP_MAXJ], om[HMOGP_MAXJ], ov[HMOGP_MAXJ];
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) {
    mu[j] = (j < J) ? m[n * J + j] : 0.0;
    vv[j] = (j < J) ? v[n * J + j] : 0.0;
    om[j] = ov[j] = 0.0;
  }
  lik_predictive<LIK>(mu, vv, param, T, lane, etab[w], om, ov);
  if (G == 1 || lane == 0)
    for (int j = 0; j < Jp; ++j) {
      mean[n * Jp + j] = om[j];
      var[n * Jp + j] = ov[j];
    }
}

// ---- Monte-Carlo log predictive density: one wave per test row, samples strided over the lanes ------------------------
//   out[n] = -log(S) + logsumexp_s log p(y_n | f_s),  f_s ~ N(m_n, diag v_n)       (e.g. bernoulli.py:130-144)
template <int LIK>
__global__ __launch_bounds__(256) void log_predictive_kernel(int J, double param, long long N, int S, unsigned long long seed,
                                                             const double* __restrict__ y, const double* __restrict__ m,
                                                             const double* __restrict__ v, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  double mu[HMOGP_MAXJ], sd[HMOGP_MAXJ];
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) {
    mu[j] = (j < J) ? m[n * J + j] : 0.0;
    sd[j] = (j < J) ? sqrt(v[n * J + j]) : 0.0;
  }
  const double yy = y[n], yaux = (LIK == HMOGP_LIK_POISSON) ? lgamma(yy + 1.0) : 0.0;
  double mx = -INFINITY, se = 0.0;  // running max / sum of exp(l - max)
  for (int s = lane; s < S; s += 64) {
    double f[HMOGP_MAXJ];
#pragma unroll
    for (int j = 0; j < HMOGP_MAXJ; j += 2) {
      double z0 = 0.0, z1 = 0.0;
      if (j < J) normal_pair(seed, n, s, j >> 1, z0, z1);
      f[j] = mu[j] + sd[j] * z0;
      if (j + 1 < HMOGP_MAXJ) f[j + 1] = mu[j + 1] + sd[j + 1] * z1;
    }
    const double l = lik_logpdf_sample<LIK>(yy, yaux, f, param);
    if (l > mx) {
      se = se * exp(mx - l) + 1.0;
      mx = l;
    } else {
      se += exp(l - mx);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {  // combine (max, sumexp) pairs across the wave
    const double omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
    const double nm = fmax(mx, omx);
    se = (mx == -INFINITY ? 0.0 : se * exp(mx - nm)) + (omx == -INFINITY ? 0.0 : ose * exp(omx - nm));
    mx = nm;
  }
  if (lane == 0) out[n] = -log((double)S) + mx + log(se);
}

// ---- data generation: one lane per row -------------------------------------------------------------------------------
template <int LIK>
__global__ __launch_bounds__(256) void sample_kernel(int J, double param, long long N, unsigned long long seed,
                                                     const double* __restrict__ F, double* __restrict__ Y) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  double f[HMOGP_MAXJ];
#pragma unroll
  for (int j = 0; j < HMOGP_MAXJ; ++j) f[j] = (j < J) ? F[n * J + j] : 0.0;
  RowRng g(seed, n);
  Y[n] = lik_sample<LIK>(g, f, param);
}

}  // namespace

// =============================================================================================== launchers
void launch_sample(int lik, int J, double param, long long N, unsigned long long seed, const double* F, double* Y,
                   hipStream_t s) {
  if (N <= 0) return;
  dim3 grid((unsigned)((N + 255) / 256));
#define SK(L) hipLaunchKernelGGL((sample_kernel<L>), grid, dim3(256), 0, s, J, param, N, seed, F, Y)
  switch (lik) {
    case HMOGP_LIK_GAUSSIAN: SK(HMOGP_LIK_GAUSSIAN); break;
    case HMOGP_LIK_BERNOULLI: SK(HMOGP_LIK_BERNOULLI); break;
    case HMOGP_LIK_HETGAUSSIAN: SK(HMOGP_LIK_HETGAUSSIAN); break;
    case HMOGP_LIK_CATEGORICAL: SK(HMOGP_LIK_CATEGORICAL); break;
    case HMOGP_LIK_POISSON: SK(HMOGP_LIK_POISSON); break;
    case HMOGP_LIK_EXPONENTIAL: SK(HMOGP_LIK_EXPONENTIAL); break;
    case HMOGP_LIK_GAMMA: SK(HMOGP_LIK_GAMMA); break;
    case HMOGP_LIK_BETA: SK(HMOGP_LIK_BETA); break;
    default: throw HipError{hipErrorInvalidValue, "unknown likelihood id", __FILE__, __LINE__};
  }
#undef SK
}

void launch_windows(const double* X, long long N, int P, const double* Z, int ldz, int M, double ell, int* rowwin, int* colwin,
                    unsigned char* hit, hipStream_t s) {
  if (N <= 0) return;
  const int tiles = (int)((N + 127) / 128), ncb = (M + 127) / 128;
  if (ncb > 64) throw HipError{hipErrorInvalidValue, "exact-zero windows support M <= 8192", __FILE__, __LINE__};
  hipLaunchKernelGGL(window_init_kernel, dim3(1), dim3(64), 0, s, colwin, ncb);
  DISPATCH_P(P, hipLaunchKernelGGL((window_kernel<PP>), dim3(tiles), dim3(256), 0, s, X, N, Z, ldz, M, ell, rowwin, colwin, hit,
                                   ncb));
  hipLaunchKernelGGL(window_fix_kernel, dim3(1), dim3(256), 0, s, rowwin, colwin, hit, tiles, ncb, M, N);
}

void launch_rbf(const double* X, int ldx, long long N, int P, const double* Z, int ldz, int M, double var, double ell,
                double* K, bool same, hipStream_t s, const int* rowwin, bool exact, const RbfBatch* batch) {
  if (N <= 0 || M <= 0) return;
  RbfBatch bt = batch ? *batch : RbfBatch{};
  dim3 grid((unsigned)((N + RBF_ROWS - 1) / RBF_ROWS), (M + 511) / 512, batch ? batch->nq : 1);
  if (exact) {
    DISPATCH_P(P, hipLaunchKernelGGL((rbf_kernel<PP, true>), grid, dim3(256), 0, s, X, ldx, N, Z, ldz, M, var, ell, K,
                                     same ? 1 : 0, rowwin, bt));
  } else {
    DISPATCH_P(P, hipLaunchKernelGGL((rbf_kernel<PP, false>), grid, dim3(256), 0, s, X, ldx, N, Z, ldz, M, var, ell, K,
                                     same ? 1 : 0, rowwin, bt));
  }
}

long long quad_blocks(int lik, long long N) { return (N * lik_lanes(lik) + 255) / 256; }

void launch_quad(const QuadArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  dim3 grid((unsigned)quad_blocks(a.lik, a.N));
#define QK(L) hipLaunchKernelGGL((quad_kernel<L>), grid, dim3(256), 0, s, a)
#define QKC(D) hipLaunchKernelGGL((quad_kernel<HMOGP_LIK_CATEGORICAL, D>), grid, dim3(256), 0, s, a)
  switch (a.lik) {
    case HMOGP_LIK_GAUSSIAN: QK(HMOGP_LIK_GAUSSIAN); break;
    case HMOGP_LIK_BERNOULLI: QK(HMOGP_LIK_BERNOULLI); break;
    case HMOGP_LIK_HETGAUSSIAN: QK(HMOGP_LIK_HETGAUSSIAN); break;
    case HMOGP_LIK_CATEGORICAL:
      switch (a.dimf) {
        case 1: QKC(1); break;
        case 2: QKC(2); break;
        case 3: QKC(3); break;
        case 4: QKC(4); break;
        case 5: QKC(5); break;
        case 6: QKC(6); break;
        case 7: QKC(7); break;
        case 8: QKC(8); break;
        default: throw HipError{hipErrorInvalidValue, "Categorical needs 2 <= K <= 9", __FILE__, __LINE__};
      }
      break;
    case HMOGP_LIK_POISSON: QK(HMOGP_LIK_POISSON); break;
    case HMOGP_LIK_EXPONENTIAL: QK(HMOGP_LIK_EXPONENTIAL); break;
    case HMOGP_LIK_GAMMA: QK(HMOGP_LIK_GAMMA); break;
    case HMOGP_LIK_BETA: QK(HMOGP_LIK_BETA); break;
    default: throw HipError{hipErrorInvalidValue, "unknown likelihood id", __FILE__, __LINE__};
  }
#undef QK
#undef QKC
}

bool quad_multi_specialised(const QuadMulti& m) {
  unsigned need = 0;
  for (int i = 0; i < m.nseg; ++i) need |= qm_bit(m.seg[i].lik, m.seg[i].dimf);
  for (unsigned mask : {QM_C1, QM_C5, QM_H4, QM_LIGHT})
    if ((need & ~mask) == 0) return true;
  return false;
}

void launch_quad_multi(const QuadMulti& m_in, hipStream_t s) {
  QuadMulti m = m_in;
  unsigned blocks = 0;
  long long part = 0;
  for (int i = 0; i < m.nseg; ++i) {
    QuadSeg& g = m.seg[i];
    if (g.lik == HMOGP_LIK_CATEGORICAL && (g.dimf < 1 || g.dimf > 8))
      throw HipError{hipErrorInvalidValue, "Categorical needs 2 <= K <= 9", __FILE__, __LINE__};
    g.blk0 = blocks, g.part0 = part;
    const long long nb = quad_blocks(g.lik, g.N);
    blocks += (unsigned)nb;
    part += nb * (2 + 2 * m.Q + g.dimf + m.Q * g.dimf);
  }
  if (blocks == 0) return;
  unsigned need = 0;
  for (int i = 0; i < m.nseg; ++i) need |= qm_bit(m.seg[i].lik, m.seg[i].dimf);
#define QML(MASK)                                                                             \
  if ((need & ~(MASK)) == 0) {                                               \
    hipLaunchKernelGGL((quad_multi_kernel<MASK>), dim3(blocks), dim3(256), 0, s, m);          \
    return;                                                                                   \
  }
  QML(QM_C1) QML(QM_C5) QML(QM_H4) QML(QM_LIGHT)
#undef QML
  // [r6] any other likelihood set (BASELINE config 4's eight families ...): one launch for the five cheap families together and one
  // per expensive family present, each over the SAME block range and segment table -- a block whose segment the instantiation does
  // not carry returns at once.  The all-inclusive instantiation this replaces is allocated for the worst body it inlines (256 VGPRs +
  // 38 AGPRs, 321 spilled SGPRs, 144 bytes of scratch, one wave per SIMD) and ran every cheap segment at that occupancy.
#define QMS(MASK)                                                                             \
  if ((need & (MASK)) != 0) hipLaunchKernelGGL((quad_multi_kernel<MASK>), dim3(blocks), dim3(256), 0, s, m);
  QMS(QM_LIGHT)
  QMS(qm_bit(HMOGP_LIK_GAMMA, 0)) QMS(qm_bit(HMOGP_LIK_BETA, 0))
  QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 1)) QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 2)) QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 3))
  QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 4)) QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 5)) QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 6))
  QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 7)) QMS(qm_bit(HMOGP_LIK_CATEGORICAL, 8))
#undef QMS
}

void launch_var_exp(int lik, int J, double param, long long N, const double* y, const double* m, const double* v, double* ve,
                    double* dm, double* dv, hipStream_t s, unsigned quirks) {
  if (N <= 0) return;
  dim3 grid((unsigned)quad_blocks(lik, N));
#define VK(L) hipLaunchKernelGGL((var_exp_kernel<L>), grid, dim3(256), 0, s, J, param, N, y, m, v, ve, dm, dv, quirks)
#define VKC(D) hipLaunchKernelGGL((var_exp_kernel<HMOGP_LIK_CATEGORICAL, D>), grid, dim3(256), 0, s, J, param, N, y, m, v, ve, dm, dv, quirks)
  switch (lik) {
    case HMOGP_LIK_GAUSSIAN: VK(HMOGP_LIK_GAUSSIAN); break;
    case HMOGP_LIK_BERNOULLI: VK(HMOGP_LIK_BERNOULLI); break;
    case HMOGP_LIK_HETGAUSSIAN: VK(HMOGP_LIK_HETGAUSSIAN); break;
    case HMOGP_LIK_CATEGORICAL:
      switch (J) {
        case 1: VKC(1); break;
        case 2: VKC(2); break;
        case 3: VKC(3); break;
        case 4: VKC(4); break;
        case 5: VKC(5); break;
        case 6: VKC(6); break;
        case 7: VKC(7); break;
        case 8: VKC(8); break;
        default: throw HipError{hipErrorInvalidValue, "Categorical needs 2 <= K <= 9", __FILE__, __LINE__};
      }
      break;
    case HMOGP_LIK_POISSON: VK(HMOGP_LIK_POISSON); break;
    case HMOGP_LIK_EXPONENTIAL: VK(HMOGP_LIK_EXPONENTIAL); break;
    case HMOGP_LIK_GAMMA: VK(HMOGP_LIK_GAMMA); break;
    case HMOGP_LIK_BETA: VK(HMOGP_LIK_BETA); break;
    default: throw HipError{hipErrorInvalidValue, "unknown likelihood id", __FILE__, __LINE__};
  }
#undef VK
#undef VKC
}

void launch_predictive(int lik, int J, int Jp, double param, int T, long long N, const double* m, const double* v,
                       double* mean, double* var, hipStream_t s) {
  if (N <= 0) return;
  dim3 grid((unsigned)((N * lik_pred_lanes(lik) + 255) / 256));
#define PK(L) hipLaunchKernelGGL((predictive_kernel<L>), grid, dim3(256), 0, s, J, Jp, param, T, N, m, v, mean, var)
  switch (lik) {
    case HMOGP_LIK_GAUSSIAN: PK(HMOGP_LIK_GAUSSIAN); break;
    case HMOGP_LIK_BERNOULLI: PK(HMOGP_LIK_BERNOULLI); break;
    case HMOGP_LIK_HETGAUSSIAN: PK(HMOGP_LIK_HETGAUSSIAN); break;
    case HMOGP_LIK_CATEGORICAL: PK(HMOGP_LIK_CATEGORICAL); break;
    case HMOGP_LIK_POISSON: PK(HMOGP_LIK_POISSON); break;
    case HMOGP_LIK_EXPONENTIAL: PK(HMOGP_LIK_EXPONENTIAL); break;
    case HMOGP_LIK_GAMMA: PK(HMOGP_LIK_GAMMA); break;
    case HMOGP_LIK_BETA: PK(HMOGP_LIK_BETA); break;
    default: throw HipError{hipErrorInvalidValue, "unknown likelihood id", __FILE__, __LINE__};
  }
#undef PK
}

void launch_log_predictive(int lik, int J, double param, long long N, int S, unsigned long long seed, const double* y,
                           const double* m, const double* v, double* out, hipStream_t s) {
  if (N <= 0) return;
  dim3 grid((unsigned)((N + 3) / 4));
#define LK(L) hipLaunchKernelGGL((log_predictive_kernel<L>), grid, dim3(256), 0, s, J, param, N, S, seed, y, m, v, out)
  switch (lik) {
    case HMOGP_LIK_GAUSSIAN: LK(HMOGP_LIK_GAUSSIAN); break;
    case HMOGP_LIK_BERNOULLI: LK(HMOGP_LIK_BERNOULLI); break;
    case HMOGP_LIK_HETGAUSSIAN: LK(HMOGP_LIK_HETGAUSSIAN); break;
    case HMOGP_LIK_CATEGORICAL: LK(HMOGP_LIK_CATEGORICAL); break;
    case HMOGP_LIK_POISSON: LK(HMOGP_LIK_POISSON); break;
    case HMOGP_LIK_EXPONENTIAL: LK(HMOGP_LIK_EXPONENTIAL); break;
    default: throw HipError{hipErrorInvalidValue, "the reference defines no log_predictive for this likelihood", __FILE__, __LINE__};
  }
#undef LK
}

void launch_colstats(const double* Kh, const double* Pt, const double* a, const double* alpha, const double* alpha0,
                     const double* beta0, const double* X, int P, const double* Z, int ldz, long long N, int M, int rows,
                     bool want_z, double* partials, hipStream_t s, const int* colwin, const ColBatch* batch, int max_blocks,
                     const double* Ar, const double* ell) {
  if (N <= 0) return;
  ColBatch bt = batch ? *batch : ColBatch{};
  dim3 grid((M + 511) / 512, (unsigned)((N + rows - 1) / rows), batch ? batch->nq : 1);
  if (max_blocks > 0) {   // cap the blocks in flight: each takes several row splits in turn
    const long long per_y = (long long)grid.x * grid.z;
    grid.y = (unsigned)std::max<long long>(1, std::min<long long>(grid.y, max_blocks / per_y));
  }
  static const bool scalar_env = [] {   // HMOGP_COLSTATS_SCALAR=0: vector loads of the row values for every P (A/B runs)
    const char* e = getenv("HMOGP_COLSTATS_SCALAR");
    return !(e && e[0] == '0');
  }();
#define HM_COLSTATS(ST, SLV)                                                                                                        \
  do {                                                                                                                            \
    if (colwin || P == 1 || !scalar_env) {   /* (P = 1 never spilled, and its vector loads are faster: C2's column statistics 10.8 vs 11.9 ms) */ \
      DISPATCH_P(P, hipLaunchKernelGGL((colstats_kernel<PP, ST, SLV, true>), grid, dim3(256), 0, s, Kh, Pt, a, alpha, alpha0, beta0, X, \
                                       Z, ldz, N, M, rows, want_z ? 1 : 0, partials, colwin, bt, Ar, ell));                     \
    } else {                                                                                                                      \
      DISPATCH_P(P, hipLaunchKernelGGL((colstats_kernel<PP, ST, SLV, false>), grid, dim3(256), 0, s, Kh, Pt, a, alpha, alpha0, beta0, X, \
                                       Z, ldz, N, M, rows, want_z ? 1 : 0, partials, colwin, bt, Ar, ell));                     \
    }                                                                                                                             \
  } while (0)
  if (Ar && ell) { HM_COLSTATS(true, true); }
  else if (Ar) { HM_COLSTATS(true, false); }
  else if (ell) { HM_COLSTATS(false, true); }
  else { HM_COLSTATS(false, false); }
#undef HM_COLSTATS
}

// dst[q * sDst] += sum_m v[q][m]   (the per-column s2 of the column statistics -> the bundle's sl_q; fixed order)
__global__ __launch_bounds__(256) void sum_cols_kernel(const double* __restrict__ v, int M, double* __restrict__ dst, long long sDst) {
  __shared__ double scratch[16];
  const double* x = v + (long long)blockIdx.x * M;
  double s = 0.0;
  for (int m = threadIdx.x; m < M; m += 256) s += x[m];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) dst[(long long)blockIdx.x * sDst] += s;
}
void launch_sum_cols(const double* v, int Q, int M, double* dst, long long sDst, hipStream_t s) {
  hipLaunchKernelGGL(sum_cols_kernel, dim3(Q), dim3(256), 0, s, v, M, dst, sDst);
}

// ---- strict q(f) (HMOGP_CFG_STRICT_QF): row statistics of the solve-based forms of svmogp_inf.py:212-218 ------------------------
// One wave per row.  phase 0:  p = A m_q,  c = rowsum(T .* T) - rowsum(A .* K^)    with A = K^ Kuu^-1, T = A L_q  ==  the reference's
// sum(square(dtrmm(L_q^T, R)), 0) - sum(R * Kfu^T, 0)  with R = dpotrs(Luu, Kfu^T) -- [r6] through X = K^ Luu^-T alone:
// p = X (Luu^-1 m), rowsum(A .* K^) = rowsum(X .* X), T = X (Luu^-1 L_q)  (the fallback of trsm_panel.hip's fused statistics).
// phase 1:  pg = K^ a,  cg = rowsum(P~ .* K^)  and their r2-weighted twins pt, ct, with P~ = A (S Kuu^-1 - I): what the reference's
// dL_dKmn (svmogp_inf.py:157-161) reduces to against K^ in svmogp.py:116-156.
template <int P>
__global__ __launch_bounds__(256) void strict_rowstats_kernel(StrictRows a) {
  const int q = blockIdx.y, lane = threadIdx.x & 63;
  const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= a.n) return;
  const int M = a.M;
  const double* kh = a.Kh + q * a.sK + n * M;
  if (a.phase == 0) {
    const double* ah = a.Ah + q * a.sK + n * M;
    double sp = 0.0, st = 0.0, sk = 0.0;
    // [r6] one-solve form: `Ah` holds X = K^ Luu^-T;  A m = X (Luu^-1 m) = X w3,  rowsum(A .* K^) = rowsum(X .* X)
    // (two-solve form, a.w3 == nullptr: `Ah` holds A itself -- p = A m, rowsum(A .* K^))
    const double* w3 = a.w3 ? a.w3 + (long long)q * M : nullptr;
    if (a.t2) {        // rowsum(T .* T) came out of the product's epilogue (gemm_rowpass.hip, fs_sq): T was never stored
      for (int m = lane; m < M; m += 64) {
        const double av = ah[m];
        sp += av * (w3 ? w3[m] : a.mu[(long long)m * a.Q + q]);
        sk += av * (w3 ? av : kh[m]);
      }
      sp = wave_sum(sp), sk = wave_sum(sk);
      if (lane == 0) a.p[q * a.ldn + n] = sp, a.c[q * a.ldn + n] = a.t2[q * a.ldn + n] - sk;
      return;
    }
    const double* tt = a.Tt + q * a.sK + n * M;
    for (int m = lane; m < M; m += 64) {
      const double av = ah[m], tv = tt[m];
      sp += av * (w3 ? w3[m] : a.mu[(long long)m * a.Q + q]);
      st += tv * tv;
      sk += av * (w3 ? av : kh[m]);
    }
    sp = wave_sum(sp), st = wave_sum(st), sk = wave_sum(sk);
    if (lane == 0) a.p[q * a.ldn + n] = sp, a.c[q * a.ldn + n] = st - sk;
    return;
  }
  const double* pt = a.Pt + q * a.sK + n * M;
  const double* av = a.a + (long long)q * M;
  const double* Z = a.Z + (long long)q * a.sZ;
  const double ell = a.ell[q];
  double xv[P];
#pragma unroll
  for (int p = 0; p < P; ++p) xv[p] = a.X[n * P + p];
  const double xsq = sumsq<P>(xv);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int m = lane; m < M; m += 64) {
    double zv[P];
#pragma unroll
    for (int p = 0; p < P; ++p) zv[p] = Z[(long long)m * a.ldz + p];
    const double r2 = rbf_r2<P>(xv, xsq, zv, sumsq<P>(zv), ell);
    const double k = kh[m], ka = k * av[m], pk = pt[m] * k;
    s0 += ka, s1 += pk, s2 += ka * r2, s3 += pk * r2;
  }
  s0 = wave_sum(s0), s1 = wave_sum(s1), s2 = wave_sum(s2), s3 = wave_sum(s3);
  if (lane == 0) {
    a.pg[q * a.ldn + n] = s0, a.cg[q * a.ldn + n] = s1;
    if (a.pt) a.pt[q * a.ldn + n] = s2, a.ct[q * a.ldn + n] = s3;
  }
}
// strict q(f): the diagonal-block step of the blocked triangular solves  V Luu^T = K^  (DIR 0, forward over the block's columns)
// and  A Luu = V  (DIR 1, backward) that make up the reference's dpotrs (svmogp_inf.py:214).  One thread = one row of the n x M
// right-hand side with its <= 32 unknowns in registers, the 32 x 32 diagonal block of Luu in LDS (broadcast reads); TRUE
// substitution, every unknown divided by its pivot as in LAPACK's dtrsm.  (A product with an explicit inverse of the block -- of
// any width down to 8 -- loses cond(block) more digits: 5e-8 instead of 3e-10 in m_fd at cond(K_uu) = 1e7, measured.)  The
// off-diagonal updates between two such steps are plain GEMMs (launch_gemm_f64, alpha = -1, beta = 1).
template <int DIR>
__global__ __launch_bounds__(256) void trsm_diag_kernel(double* __restrict__ V, long long sV, const double* __restrict__ L,
                                                        long long sL, int M, int j0, int nb, long long n, int wide, int u0, int u1,
                                                        const double* __restrict__ Vsrc) {
  // (the unknowns live in LDS, one column of 256 lanes per unknown -- conflict-free -- and the loops run at run time: with
  //  x[32] in registers and both loops unrolled the compiler hoists all 528 broadcast reads and spills ~1 KB per lane)
  __shared__ double Ls[32][33];
  __shared__ double xs[32][258];
  V += (long long)blockIdx.y * sV, L += (long long)blockIdx.y * sL;
  // Vsrc (wide launches only): this launch is the FIRST touch of its columns -- x and the updated columns are read from the matrix
  // the solve started from (same layout as V) and written to V: no copy of that matrix beforehand
  const double* Vin = Vsrc ? Vsrc + (long long)blockIdx.y * sV : V;
  const int t = threadIdx.x;
  for (int e = t; e < 32 * 32; e += 256) {
    const int r = e >> 5, c = e & 31;
    Ls[r][c] = (r < nb && c < nb) ? L[(long long)(j0 + r) * M + j0 + c] : (r == c ? 1.0 : 0.0);
  }
  const long long row0 = (long long)blockIdx.x * 256, row = row0 + t;
  const bool valid = row < n;
  double* v = V + (valid ? row : 0) * M + j0;
  // [r5] the block's 256 x 32 right-hand sides enter (and leave) through 16-byte accesses in which 16 consecutive lanes cover one
  // row's 256 bytes: one row per lane (32 loads of 8 bytes, lanes 8 KB apart) touched 64 cache lines per instruction
  // (substitution launches 47 -> 30 ms per strict step at the headline size: DESIGN 12g)
  // (`wide` is decided by the launcher: nb == 32, even M / j0 / batch stride, 16-byte aligned V)
  if (wide) {
#pragma unroll 4
    for (int e = t; e < 256 * 16; e += 256) {
      const int r = e >> 4, ch = e & 15;
      f64x2 v2 = f64x2{0.0, 0.0};
      if (row0 + r < n) v2 = *reinterpret_cast<const f64x2*>(Vin + (row0 + r) * M + j0 + 2 * ch);
      xs[2 * ch][r] = v2.x, xs[2 * ch + 1][r] = v2.y;
    }
  } else {
    for (int c = 0; c < 32; ++c) xs[c][t] = (valid && c < nb) ? v[c] : 0.0;
  }
  __syncthreads();
  // Chunks of 8 unknowns: their right-hand sides sit in 8 registers while the contributions of the unknowns solved before are
  // subtracted (one LDS read of the unknown + 8 broadcast reads of L feed 8 INDEPENDENT fused multiply-adds: the first version
  // ran one dependent chain per unknown, 25 ms per step at 200 000 rows), then the 8 x 8 triangle is solved in registers.
  // Padded rows / columns of the last block are identity: harmless.
  if (DIR == 0) {
    for (int c0 = 0; c0 < 32; c0 += 8) {
      if (c0 >= nb) break;
      double acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = xs[c0 + r][t];
      for (int i = 0; i < c0; ++i) {
        const double xi = xs[i][t];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] -= xi * Ls[c0 + r][i];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int q = 0; q < r; ++q) acc[r] -= acc[q] * Ls[c0 + r][c0 + q];
        acc[r] = acc[r] / Ls[c0 + r][c0 + r];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) xs[c0 + r][t] = acc[r];
    }
  } else {
    for (int c0 = 24; c0 >= 0; c0 -= 8) {
      if (c0 >= nb) continue;
      double acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = xs[c0 + r][t];
      for (int i = 31; i >= c0 + 8; --i) {
        const double xi = xs[i][t];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] -= xi * Ls[i][c0 + r];
      }
#pragma unroll
      for (int r = 7; r >= 0; --r) {
#pragma unroll
        for (int q = 7; q > r; --q) acc[r] -= acc[q] * Ls[c0 + q][c0 + r];
        acc[r] = acc[r] / Ls[c0 + r][c0 + r];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) xs[c0 + r][t] = acc[r];
    }
  }
  if (wide) {
    __syncthreads();
#pragma unroll 4
    for (int e = t; e < 256 * 16; e += 256) {
      const int r = e >> 4, ch = e & 15;
      if (row0 + r < n) *reinterpret_cast<f64x2*>(V + (row0 + r) * M + j0 + 2 * ch) = f64x2{xs[2 * ch][r], xs[2 * ch + 1][r]};
    }
    // [r5] The RIGHT-LOOKING update of the columns [u0, u1) that this block's unknowns still feed inside the current 128-column
    // block of the two-level scheme -- V[:, c] -= sum_k x[:, j0 + k] L[c][j0 + k] (DIR 0) / L[j0 + k][c] (DIR 1) -- on the
    // matrix cores, 32 columns at a time, with the freshly solved x read from LDS where it already sits: the separate
    // 32-column GEMM updates (one quarter of a 128-wide tile used, A and C streamed from HBM again) cost as much as the
    // 128-column updates that carry 12 x their flops.  A wave owns 64 rows; D fragment: col = lane & 15, row = (lane >> 4) + 4 reg.
    const int lane = t & 63, w = t >> 6, lr = lane & 15, lk = lane >> 4;
    for (int g0 = u0; g0 < u1; g0 += 32) {
      __syncthreads();                      // Ls: the substitution / the previous group's products are done with it
      for (int e = t; e < 32 * 32; e += 256) {
        if (DIR == 0) {
          const int c = e >> 5, k = e & 31;
          Ls[k][c] = L[(long long)(g0 + c) * M + j0 + k];
        } else {
          const int k = e >> 5, c = e & 31;
          Ls[k][c] = L[(long long)(j0 + k) * M + g0 + c];
        }
      }
      f64x4 acc[4][2];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long rr = row0 + w * 64 + a * 16 + 4 * r + lk;
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b][r] = rr < n ? Vin[rr * M + g0 + b * 16 + lr] : 0.0;
        }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        double fa[4], fb[2];
#pragma unroll
        for (int a = 0; a < 4; ++a) fa[a] = -xs[kk * 4 + lk][w * 64 + a * 16 + lr];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = Ls[kk * 4 + lk][b * 16 + lr];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long rr = row0 + w * 64 + a * 16 + 4 * r + lk;
          if (rr >= n) continue;
#pragma unroll
          for (int b = 0; b < 2; ++b) V[rr * M + g0 + b * 16 + lr] = acc[a][b][r];
        }
    }
    return;
  }
  if (valid)
    for (int c = 0; c < nb; ++c) v[c] = xs[c][t];
}
bool trsm_diag_can_fuse(const double* V, long long sV, int M) {
  return (M % 32) == 0 && (sV & 1) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0;
}
void launch_trsm_diag(int dir, double* V, long long sV, const double* L, long long sL, int M, int j0, int nb, long long n, int Q,
                      hipStream_t s, int u0, int u1, const double* Vsrc) {
  if (n <= 0 || nb <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256), Q);
  const int wide = (nb == 32 && (M & 1) == 0 && (j0 & 1) == 0 && (sV & 1) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0) ? 1 : 0;
  if (u1 > u0 && (!wide || ((u1 - u0) & 31) || (u0 & 31)))
    throw HipError{hipErrorInvalidValue, "trsm_diag: fused update needs full 32-column blocks and aligned rows", __FILE__, __LINE__};
  if (Vsrc && (!wide || (reinterpret_cast<uintptr_t>(Vsrc) & 15)))
    throw HipError{hipErrorInvalidValue, "trsm_diag: a first-touch source needs the row-coalesced path", __FILE__, __LINE__};
  if (dir == 0) hipLaunchKernelGGL((trsm_diag_kernel<0>), grid, dim3(256), 0, s, V, sV, L, sL, M, j0, nb, n, wide, u0, u1, Vsrc);
  else hipLaunchKernelGGL((trsm_diag_kernel<1>), grid, dim3(256), 0, s, V, sV, L, sL, M, j0, nb, n, wide, u0, u1, Vsrc);
}
void launch_strict_rowstats(const StrictRows& a, hipStream_t s) {
  if (a.n <= 0) return;
  dim3 grid((unsigned)((a.n + 3) / 4), a.Q);
  DISPATCH_P(a.P, hipLaunchKernelGGL((strict_rowstats_kernel<PP>), grid, dim3(256), 0, s, a));
}

namespace {
__global__ __launch_bounds__(256) void reduce_rows_multi_kernel(SmallQuadRed qr, double* __restrict__ dst) {
  __shared__ double scratch[16];
  const int k = blockIdx.x;
  for (int sg = 0; sg < qr.nseg; ++sg) {
    const auto& g = qr.s[sg];
    if (k >= g.nscal) continue;              // (block-uniform)
    double s = 0.0;
    for (long long b = threadIdx.x; b < g.nrows; b += blockDim.x) s += g.part[b * g.nscal + k];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) dst[g.off[k]] += s;
    __syncthreads();
  }
}
}  // namespace
void launch_reduce_rows_multi(const SmallQuadRed& qr, double* dst, hipStream_t s) {
  int len = 0;
  for (int i = 0; i < qr.nseg; ++i) len = std::max(len, qr.s[i].nscal);
  if (len <= 0) return;
  hipLaunchKernelGGL(reduce_rows_multi_kernel, dim3(len), dim3(256), 0, s, qr, dst);
}

void launch_reduce_rows(const double* partials, long long nrows, int len, const long long* off, double* dst, bool accumulate,
                        hipStream_t s) {
  if (len <= 0) return;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(len), dim3(256), 0, s, partials, nrows, len, off, dst, accumulate ? 1 : 0);
}

void launch_reduce_slabs(const double* slabs, int nslabs, long long stride, long long len, double* dst, bool accumulate,
                         hipStream_t s, int nb, long long sSlabs, long long sDst) {
  if (len <= 0) return;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((len + 15) / 16), nb), dim3(256), 0, s, slabs, nslabs, stride, len,
                     dst, accumulate ? 1 : 0, sSlabs, sDst);
}

void launch_reduce_slabs_lower(const double* slabs, int nslabs, int M, double* dst, bool accumulate, hipStream_t s, int nb,
                               long long sSlabs, long long sDst) {
  const int tiles = (M + 127) / 128;
  hipLaunchKernelGGL(reduce_slabs_lower_kernel, dim3(tiles * (tiles + 1) / 2, 32, nb), dim3(256), 0, s, slabs, nslabs, M, dst,
                     accumulate ? 1 : 0, sSlabs, sDst);
}

void launch_mirror_lower(double* A, int Q, int M, long long stride, hipStream_t s) {
  hipLaunchKernelGGL(mirror_lower_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, A, M, stride);
}

void launch_wire_copy(double* bundle, double* wire, long long NG, int Q, int M, long long per_q, int dir, hipStream_t s) {
  const long long MM = (long long)M * M, Mtri = (long long)M * (M + 1) / 2, wire_per_q = Mtri + (per_q - MM);
  hipLaunchKernelGGL(wire_tri_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, bundle + NG, wire + NG, M, per_q, wire_per_q,
                     dir);
  const long long rest = NG + (per_q - MM) * Q;
  hipLaunchKernelGGL(wire_rest_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, s, bundle, wire, NG, MM, Mtri, per_q,
                     wire_per_q, Q, dir);
}

void launch_raw_kmn(const double* a, const double* gm, const double* gv, int J, int j, double w, const double* Pt, int M,
                    long long N, double* out, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(raw_kmn_kernel, dim3((unsigned)((N + 255) / 256), M), dim3(256), 0, s, a, gm, gv, J, j, w, Pt, M, N, out);
}

void launch_gammaln1p(const double* y, double* out, long long N, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(gammaln1p_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, y, out, N);
}
