// post.hip -- element-wise / reduction kernels of the replicated M x M algebra (gfx950): the pieces of
// SVMOGPInf.calculate_KL (svmogp_inf.py:227-250), calculate_gradients (:111-183) and of the K_uu half of
// SVMOGP.parameters_changed (svmogp.py:116,154) that are not GEMMs.
#include <algorithm>

#include "post.h"
#include "rbf_device.h"

namespace {

// dst[q] = src[q] + jit[q] * I     (GPy jitchol adds the jitter to the factorised copy only, util.py:198)
__global__ void add_diag_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int M,
                                     const double* __restrict__ jit) {
  const int q = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= M) return;
  const long long i = ((long long)q * M + r) * M + c;
  dst[i] = src[i] + ((r == c) ? jit[q] : 0.0);
}

__global__ void sub_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) C[i] = A[i] - B[i];
}

__global__ void identity_kernel(double* __restrict__ A, int M) {
  const long long o = (long long)blockIdx.z * M * M;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < M) A[o + (long long)r * M + c] = (r == c) ? 1.0 : 0.0;
}

// strict q(f): D = (Kuu^-1 S)^T - I = S Kuu^-1 - I, the right factor of P~ = A D (svmogp_inf.py:157-159: tmp = 2 (S Kuui - I))
__global__ void strict_d_kernel(const double* __restrict__ KiS, double* __restrict__ D, int M) {
  const long long o = (long long)blockIdx.z * M * M;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= M) return;
  D[o + (long long)r * M + c] = KiS[o + (long long)c * M + r] - (r == c ? 1.0 : 0.0);
}

// [r6] strict q(f), one-solve form.  With X = K^ Luu^-T (the FORWARD substitution of the reference's dpotrs, svmogp_inf.py:214) the
// second half of that solve moves from the n x M side onto the M x M side:
//   A m = X (Luu^-1 m),   A L_q = X (Luu^-1 L_q),   A (S Kuu^-1 - I) = X (Luu^-1 (S Kuu^-1 - I))
// The three right factors come out of ONE forward row-solve V Luu^T = [ L_q^T ; Kuu^-1 S - I ; m^T ]  (2 M + 1 rows):
//   rows [0, M)  -> (Luu^-1 L_q)^T,  rows [M, 2 M) -> (Luu^-1 (S Kuu^-1 - I))^T,  row 2 M -> (Luu^-1 m)^T.
// stack: V[q][i][j] = L[q][j][i] | V[q][M + i][j] = KiS[q][i][j] - (i == j) | V[q][2 M][j] = mu[j][q]     (32 x 32 LDS tiles)
__global__ __launch_bounds__(256) void strict_stack_kernel(const double* __restrict__ L, const double* __restrict__ KiS,
                                                           const double* __restrict__ mu, double* __restrict__ V, long long sV,
                                                           int M, int Q) {
  __shared__ double tile[32][33];
  const int q = blockIdx.z, bi = blockIdx.y * 32, bj = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long MM = (long long)M * M;
  double* Vq = V + q * sV;
  for (int r = ty; r < 32; r += 8)                                    // block 0: transpose of L_q
    tile[r][tx] = (bj + r < M && bi + tx < M) ? L[q * MM + (long long)(bj + r) * M + bi + tx] : 0.0;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bi + r < M && bj + tx < M) Vq[(long long)(bi + r) * M + bj + tx] = tile[tx][r];
  for (int r = ty; r < 32; r += 8)                                    // block 1: Kuu^-1 S - I
    if (bi + r < M && bj + tx < M)
      Vq[(long long)(M + bi + r) * M + bj + tx] = KiS[q * MM + (long long)(bi + r) * M + bj + tx] - (bi + r == bj + tx ? 1.0 : 0.0);
  if (blockIdx.y == 0 && ty == 0 && bj + tx < M) Vq[2LL * M * M + bj + tx] = mu[(long long)(bj + tx) * Q + q];
}
// unstack: W[q][i][j] = V[q][j][i],  W2[q][i][j] = V[q][M + j][i],  w3[q][j] = V[q][2 M][j]
__global__ __launch_bounds__(256) void strict_unstack_kernel(const double* __restrict__ V, long long sV, double* __restrict__ W,
                                                             double* __restrict__ W2, double* __restrict__ w3, int M) {
  __shared__ double tile[2][32][33];
  const int q = blockIdx.z, bi = blockIdx.y * 32, bj = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long MM = (long long)M * M;
  const double* Vq = V + q * sV;
  for (int r = ty; r < 32; r += 8) {
    const bool in = bj + r < M && bi + tx < M;
    tile[0][r][tx] = in ? Vq[(long long)(bj + r) * M + bi + tx] : 0.0;
    tile[1][r][tx] = in ? Vq[(long long)(M + bj + r) * M + bi + tx] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bi + r < M && bj + tx < M) {
      W[q * MM + (long long)(bi + r) * M + bj + tx] = tile[0][tx][r];
      W2[q * MM + (long long)(bi + r) * M + bj + tx] = tile[1][tx][r];
    }
  if (blockIdx.y == 0 && ty == 0 && bj + tx < M) w3[(long long)q * M + bj + tx] = Vq[2LL * M * M + bj + tx];
}
// B[q][j][i] = A[q][i][j]   (batched M x M transpose, strides sA / sB)
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ A, long long sA, double* __restrict__ B, long long sB,
                                                        int M) {
  __shared__ double tile[32][33];
  const int q = blockIdx.z, bi = blockIdx.y * 32, bj = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = (bi + r < M && bj + tx < M) ? A[q * sA + (long long)(bi + r) * M + bj + tx] : 0.0;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bj + r < M && bi + tx < M) B[q * sB + (long long)(bj + r) * M + bi + tx] = tile[tx][r];
}

// out[q] = variance_q * max_i (K_uu^-1)_ii  -- the condition estimate of hmogp_outputs.cond_est, early (strict q(f): which form)
__global__ __launch_bounds__(256) void cond_probe_kernel(const double* __restrict__ Kuui, const double* __restrict__ var, int M,
                                                         double* __restrict__ out) {
  __shared__ double sm[4];
  const int q = blockIdx.x, t = threadIdx.x;
  double k = 0.0;
  for (int i = t; i < M; i += 256) k = fmax(k, Kuui[(long long)q * M * M + (long long)i * M + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) k = fmax(k, __shfl_xor(k, o, 64));
  if ((t & 63) == 0) sm[t >> 6] = k;
  __syncthreads();
  if (t == 0) out[q] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3])) * var[q];
}

// T = tril(C) + tril(C^T, -1): the lower-triangular matrix with x^T T x == x^T C x (C need not be exactly symmetric).
// With it the quadratic forms k^T C k of the forward contraction cost half the products (GemmArgs::b_tri).
__global__ void tri_fold_kernel(const double* __restrict__ C, double* __restrict__ T, int M) {
  const long long o = (long long)blockIdx.z * M * M;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= M) return;
  double v = 0.0;
  if (c == r) v = C[o + (long long)r * M + c];
  else if (c < r) v = C[o + (long long)r * M + c] + C[o + (long long)c * M + r];
  T[o + (long long)r * M + c] = v;
}

// out[q][0] = sum(Kuui .* S), [1] = m^T a, [2] = sum log|diag Luu|, [3] = sum log|diag L|, [4] = #inf in Sqi
// [r5] behind the Q * KL_BLOCKS * 5 partials: [Q][KL_BLOCKS] block maxima of diag(Kuui) -- variance * max_i (K_uu^-1)_ii is a lower
// bound of cond(K_uu) (lambda_max(K^-1) >= its largest diagonal entry, lambda_max(K) >= variance), within 30-150x of it on RBF
// matrices: what hmogp_outputs.cond_est / HMOGP_FLAG_ILL_CONDITIONED report.
__global__ __launch_bounds__(256) void kl_terms_kernel(const double* __restrict__ Kuui, const double* __restrict__ S,
                                                       const double* __restrict__ m_u, const double* __restrict__ a,
                                                       const double* __restrict__ Luu, const double* __restrict__ L,
                                                       const double* __restrict__ Sqi, int Q, int M,
                                                       double* __restrict__ out) {
  __shared__ double scratch[16];
  const int q = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
  const long long MM = (long long)M * M, base = (long long)q * MM;
  double tr = 0.0, ninf = 0.0;
  for (long long i = (long long)b * 256 + t; i < MM; i += 256LL * KL_BLOCKS) {
    tr += Kuui[base + i] * S[base + i];
    if (Sqi) ninf += isinf(Sqi[base + i]) ? 1.0 : 0.0;
  }
  double ma = 0.0, l1 = 0.0, l2 = 0.0, kmax = 0.0;
  for (int i = b * 256 + t; i < M; i += 256 * KL_BLOCKS) {
    kmax = fmax(kmax, Kuui[base + (long long)i * M + i]);
    ma += m_u[(long long)i * Q + q] * a[(long long)q * M + i];
    l1 += log(fabs(Luu[base + (long long)i * M + i]));
    l2 += log(fabs(L[base + (long long)i * M + i]));
  }
  tr = block_sum(tr, scratch);
  ma = block_sum(ma, scratch);
  l1 = block_sum(l1, scratch);
  l2 = block_sum(l2, scratch);
  ninf = block_sum(ninf, scratch);
  __shared__ double smax[4];
  for (int o2 = 32; o2; o2 >>= 1) kmax = fmax(kmax, __shfl_xor(kmax, o2, 64));
  if ((t & 63) == 0) smax[t >> 6] = kmax;
  __syncthreads();
  if (t == 0) {
    double* o = out + ((long long)q * KL_BLOCKS + b) * 5;
    o[0] = tr, o[1] = ma, o[2] = l1, o[3] = l2, o[4] = ninf;
    out[(long long)Q * KL_BLOCKS * 5 + (long long)q * KL_BLOCKS + b] = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
  }
}

// dL_dKmm (svmogp_inf.py:130-133,151-154,166,170):
//   X = G - GSK - GSK^T - (Kuui r) a^T ;  dVE = (X + X^T)/2 ;  dKL = Kuui/2 - KSK/2 - a a^T/2 ;  out = dVE - dKL
__global__ void dkmm_kernel(const double* __restrict__ G, const double* __restrict__ GSK, const double* __restrict__ Kuui,
                            const double* __restrict__ KSK, const double* __restrict__ Kr, const double* __restrict__ a,
                            double* __restrict__ out, int M) {
  const int q = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const long long b = (long long)q * M * M, ij = b + (long long)i * M + j, ji = b + (long long)j * M + i;
  const double* kr = Kr + (long long)q * M;
  const double* av = a + (long long)q * M;
  const double xij = G[ij] - GSK[ij] - GSK[ji] - kr[i] * av[j];
  const double xji = G[ji] - GSK[ji] - GSK[ij] - kr[j] * av[i];
  const double dve = 0.5 * (xij + xji);
  const double dkl = 0.5 * Kuui[ij] - 0.5 * KSK[ij] - 0.5 * (av[i] * av[j]);
  out[ij] = dve - dkl;
}

// The same through 32 x 32 tiles staged in LDS: the transposed entries G[j][i], GSK[j][i] come from the mirrored tile, read row-wise
// (the element-wise kernel above reads them column-wise -- uncoalesced 8-byte loads: 245 us at M = 1024, Q = 3, on the
// critical path of the gradient tail).  Same operations in the same order per element: bit-identical results.
__global__ __launch_bounds__(256) void dkmm_tiled_kernel(const double* __restrict__ G, const double* __restrict__ GSK,
                                                         const double* __restrict__ Kuui, const double* __restrict__ KSK,
                                                         const double* __restrict__ Kr, const double* __restrict__ a,
                                                         double* __restrict__ out, int M) {
  __shared__ double tG[32][33], tS[32][33];
  const int q = blockIdx.z, bi = blockIdx.y * 32, bj = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long b = (long long)q * M * M;
  for (int r = ty; r < 32; r += 8) {          // mirrored tile (rows bj.., columns bi..): [r][c] = X[bj + r][bi + c]
    const int row = bj + r, col = bi + tx;
    const bool ok = row < M && col < M;
    tG[r][tx] = ok ? G[b + (long long)row * M + col] : 0.0;
    tS[r][tx] = ok ? GSK[b + (long long)row * M + col] : 0.0;
  }
  __syncthreads();
  const double* kr = Kr + (long long)q * M;
  const double* av = a + (long long)q * M;
  for (int r = ty; r < 32; r += 8) {
    const int i = bi + r, j = bj + tx;
    if (i >= M || j >= M) continue;
    const long long ij = b + (long long)i * M + j;
    const double gij = G[ij], sij = GSK[ij], gji = tG[tx][r], sji = tS[tx][r];
    const double xij = gij - sij - sji - kr[i] * av[j];
    const double xji = gji - sji - sij - kr[j] * av[i];
    const double dve = 0.5 * (xij + xji);
    const double dkl = 0.5 * Kuui[ij] - 0.5 * KSK[ij] - 0.5 * (av[i] * av[j]);
    out[ij] = dve - dkl;
  }
}

// dL_dS = G - (Kuui - Sqi)/2        (svmogp_inf.py:131,169)
__global__ void dlds_kernel(const double* __restrict__ G, const double* __restrict__ Kuui, const double* __restrict__ Sqi,
                            double* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = G[i] - 0.5 * (Kuui[i] - Sqi[i]);
}

// gL[(r(r+1)/2 + c) * Q + q] = 2 * T[q][r][c], c <= r      (svmogp_inf.py:175-178, GPy triang_to_flat)
__global__ void pack_gl_kernel(const double* __restrict__ T, double* __restrict__ gL, int Q, int M) {
  const int q = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > r || c >= M) return;
  gL[((long long)r * (r + 1) / 2 + c) * Q + q] = 2.0 * T[((long long)q * M + r) * M + c];
}

// g_m_u[m*Q + q] = Kr[q][m] - a[q][m]     (svmogp_inf.py:130,144,168)
__global__ void gmu_kernel(const double* __restrict__ Kr, const double* __restrict__ a, double* __restrict__ g, int Q, int M) {
  const int q = blockIdx.y, m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < M) g[(long long)m * Q + q] = Kr[(long long)q * M + m] - a[(long long)q * M + m];
}

// K_uu half of the kernel-hyper / Z gradients (GPy RBF.update_gradients_full(dL_dKmm, Zq) and
// gradients_X(dL_dKmm, Zq) with X2=None: diagonal distance forced to 0, dL_dK + dL_dK^T).  One wave per row m:
//   rowout[q][m] = { sum_j EK_mj , sum_j EK_mj r2_mj , sum_j (EK_mj + EK_jm)(z_j - z_m)[p] ... }   EK = dKmm .* Kzz
template <int P>
__global__ __launch_bounds__(256) void kzz_rows_kernel(const double* __restrict__ dKmm, const double* __restrict__ Z, int ldz,
                                                       const double* __restrict__ var, const double* __restrict__ ell,
                                                       int M, double* __restrict__ rowout) {
  const int q = blockIdx.y, m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  const double* Zq = Z + (long long)q * P;
  const double* D = dKmm + (long long)q * M * M;
  const double v = var[q], l = ell[q];
  double zm[P];
#pragma unroll
  for (int p = 0; p < P; ++p) zm[p] = Zq[(long long)m * ldz + p];
  const double zmsq = sumsq<P>(zm);
  double s1 = 0.0, s2 = 0.0, gz[P];
#pragma unroll
  for (int p = 0; p < P; ++p) gz[p] = 0.0;
  for (int j = lane; j < M; j += 64) {
    double zj[P];
#pragma unroll
    for (int p = 0; p < P; ++p) zj[p] = Zq[(long long)j * ldz + p];
    double r2 = rbf_r2<P>(zm, zmsq, zj, sumsq<P>(zj), l);
    if (j == m) r2 = 0.0;
    const double kz = v * exp(-0.5 * r2);
    const double ek = D[(long long)m * M + j] * kz, ekt = D[(long long)j * M + m] * kz;
    s1 += ek;
    s2 += ek * r2;
#pragma unroll
    for (int p = 0; p < P; ++p) gz[p] += (r2 != 0.0) ? (ek + ekt) * (zj[p] - zm[p]) : 0.0;   // (quirk Q10: GPy gradients_X drops r == 0)
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
#pragma unroll
  for (int p = 0; p < P; ++p) gz[p] = wave_sum(gz[p]);
  if (lane == 0) {
    double* o = rowout + ((long long)q * M + m) * (2 + P);
    o[0] = s1;
    o[1] = s2;
#pragma unroll
    for (int p = 0; p < P; ++p) o[2 + p] = gz[p];
  }
}

// q(f_d) at arbitrary inputs (svmogp_inf.py:212-218): m[n][d] = sum_q W p_q ; v[n][d] = sum_q (B sigma2 + W^2 c_q)
__global__ void qf_combine_kernel(const double* __restrict__ p, const double* __restrict__ c, long long ldn, long long N,
                                  int Q, int Df, const double* __restrict__ W, const double* __restrict__ kappa,
                                  const double* __restrict__ var, double* __restrict__ m, double* __restrict__ v) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  for (int d = 0; d < Df; ++d) {
    double mm = 0.0, vv = 0.0;
    for (int q = 0; q < Q; ++q) {
      const double w = W[q * Df + d];
      mm += w * p[q * ldn + n];
      vv += (w * w + kappa[q * Df + d]) * var[q] + w * w * c[q * ldn + n];
    }
    m[n * Df + d] = mm;
    v[n * Df + d] = vv;
  }
}

// natural-gradient precision update: out = sym(Sqi - 2 gamma dLdS)            
// rev != 0 writes J Lambda J (rows and columns reversed): its lower Cholesky factor R gives Lambda = U U^T with U = J R J
// UPPER triangular, so that L = U^-T is the lower Cholesky factor of S = Lambda^-1 -- one factorisation instead of two
__global__ void natgrad_prec_kernel(const double* __restrict__ Sqi, const double* __restrict__ dLdS, double gamma,
                                    double* __restrict__ out, int M, int rev) {
  const int q = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const long long b = (long long)q * M * M, ij = b + (long long)i * M + j, ji = b + (long long)j * M + i;
  const double v = 0.5 * ((Sqi[ij] - 2.0 * gamma * dLdS[ij]) + (Sqi[ji] - 2.0 * gamma * dLdS[ji]));
  out[rev ? b + (long long)(M - 1 - i) * M + (M - 1 - j) : ij] = v;
}
// L[q][i][j] = T[q][M-1-j][M-1-i]  (transpose about the anti-diagonal; 32 x 32 tiles through LDS, coalesced both ways)
__global__ __launch_bounds__(256) void antitranspose_kernel(const double* __restrict__ T, double* __restrict__ L, int M) {
  __shared__ double tile[32][33];
  const int q = blockIdx.z, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* t = T + (long long)q * M * M;
  double* l = L + (long long)q * M * M;
  const int sr0 = blockIdx.y * 32, sc0 = blockIdx.x * 32;  // source tile
  for (int k = ty; k < 32; k += 8) {
    const int r = sr0 + k, c = sc0 + tx;
    tile[k][tx] = (r < M && c < M) ? t[(long long)r * M + c] : 0.0;
  }
  __syncthreads();
  // source (r, c) lands at (M-1-c, M-1-r): destination rows run over the tile's columns, reversed
  for (int k = ty; k < 32; k += 8) {
    const int c = sc0 + k, r = sr0 + (31 - tx);  // consecutive tx -> consecutive destination columns M-1-r
    if (r < M && c < M) l[(long long)(M - 1 - c) * M + (M - 1 - r)] = tile[31 - tx][k];
  }
}
// y[q][j] = sum_i A[q][i][j] x[q][i]   (A^T x: 64 columns per block, a wave per 1/16 of the rows -- coalesced across the
// wave -- and a fixed-order LDS reduction, so the result is reproducible)
__global__ __launch_bounds__(1024) void gemv_t_kernel(const double* __restrict__ A, const double* __restrict__ x,
                                                      double* __restrict__ y, int M) {
  __shared__ double part[16][64];
  const int q = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = blockIdx.x * 64 + lane;
  const double* a = A + (long long)q * M * M;
  const double* xv = x + (long long)q * M;
  double s = 0.0;
  if (j < M)
    for (int i = w; i < M; i += 16) s += a[(long long)i * M + j] * xv[i];
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && j < M) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += part[k][lane];
    y[(long long)q * M + j] = t;
  }
}
// theta1[q][i] = t1[q][i] + gamma * (g_m[i*Q+q] - 2 t2[q][i])       with t1 = Sqi m, t2 = dLdS m
__global__ void natgrad_theta1_kernel(const double* __restrict__ t1, const double* __restrict__ t2,
                                      const double* __restrict__ gm, double gamma, double* __restrict__ out, int Q, int M) {
  const int q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[(long long)q * M + i] = t1[(long long)q * M + i] + gamma * (gm[(long long)i * Q + q] - 2.0 * t2[(long long)q * M + i]);
}
// out[(r(r+1)/2 + c) * Q + q] = scale * T[q][r][c]
__global__ void pack_tril_kernel(const double* __restrict__ T, double* __restrict__ out, int Q, int M, double scale) {
  const int q = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > r || c >= M) return;
  out[((long long)r * (r + 1) / 2 + c) * Q + q] = scale * T[((long long)q * M + r) * M + c];
}
// out[i*Q + q] = v[q][i]
__global__ void scatter_mq_kernel(const double* __restrict__ v, double* __restrict__ out, int Q, int M) {
  const int q = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) out[(long long)i * Q + q] = v[(long long)q * M + i];
}

}  // namespace

void launch_natgrad_prec(const double* Sqi, const double* dLdS, double gamma, double* out, int Q, int M, bool reversed,
                         hipStream_t s) {
  hipLaunchKernelGGL(natgrad_prec_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, Sqi, dLdS, gamma, out, M, reversed ? 1 : 0);
}
void launch_antitranspose(const double* T, double* L, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(antitranspose_kernel, dim3((M + 31) / 32, (M + 31) / 32, Q), dim3(256), 0, s, T, L, M);
}
void launch_gemv_t_batched(const double* A, const double* x, double* y, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(gemv_t_kernel, dim3((M + 63) / 64, Q), dim3(1024), 0, s, A, x, y, M);
}
void launch_natgrad_theta1(const double* t1, const double* t2, const double* gm, double gamma, double* out, int Q, int M,
                           hipStream_t s) {
  hipLaunchKernelGGL(natgrad_theta1_kernel, dim3((M + 255) / 256, Q), dim3(256), 0, s, t1, t2, gm, gamma, out, Q, M);
}
void launch_pack_tril(const double* T, double* out, int Q, int M, double scale, hipStream_t s) {
  hipLaunchKernelGGL(pack_tril_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, T, out, Q, M, scale);
}
// dst1 <- src1, dst2 <- src2 unless one of the Q info words is non-zero (hmogp_qu_natgrad_async: a step that left the
// positive-definite cone commits nothing -- decided on the device, the host looks at the info words later)
__global__ __launch_bounds__(256) void commit_if_ok_kernel(const int* __restrict__ info, int Q, const double* __restrict__ src1,
                                                           double* __restrict__ dst1, long long n1, const double* __restrict__ src2,
                                                           double* __restrict__ dst2, long long n2) {
  for (int q = 0; q < Q; ++q)
    if (info[q] != 0) return;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += stride) {
    if (i < n1) dst1[i] = src1[i];
    else dst2[i - n1] = src2[i - n1];
  }
}
void launch_commit_if_ok(const int* info, int Q, const double* src1, double* dst1, long long n1, const double* src2, double* dst2,
                         long long n2, hipStream_t s) {
  const long long n = n1 + n2;
  if (n <= 0) return;
  hipLaunchKernelGGL(commit_if_ok_kernel, dim3((unsigned)std::min<long long>(2048, (n + 255) / 256)), dim3(256), 0, s, info, Q,
                     src1, dst1, n1, src2, dst2, n2);
}
void launch_scatter_mq(const double* v, double* out, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(scatter_mq_kernel, dim3((M + 255) / 256, Q), dim3(256), 0, s, v, out, Q, M);
}

void launch_add_diag_copy(const double* src, double* dst, int Q, int M, const double* d_jit, hipStream_t s) {
  hipLaunchKernelGGL(add_diag_copy_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, src, dst, M, d_jit);
}
void launch_sub(const double* A, const double* B, double* C, long long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(sub_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, A, B, C, n);
}
void launch_identity(double* A, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(identity_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, A, M);
}
void launch_strict_d(const double* KiS, double* D, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(strict_d_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, KiS, D, M);
}
void launch_cond_probe(const double* Kuui, const double* var, int Q, int M, double* out, hipStream_t s) {
  hipLaunchKernelGGL(cond_probe_kernel, dim3(Q), dim3(256), 0, s, Kuui, var, M, out);
}
void launch_strict_stack(const double* L, const double* KiS, const double* mu, double* V, long long sV, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(strict_stack_kernel, dim3((M + 31) / 32, (M + 31) / 32, Q), dim3(256), 0, s, L, KiS, mu, V, sV, M, Q);
}
void launch_strict_unstack(const double* V, long long sV, double* W, double* W2, double* w3, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(strict_unstack_kernel, dim3((M + 31) / 32, (M + 31) / 32, Q), dim3(256), 0, s, V, sV, W, W2, w3, M);
}
void launch_transpose_batched(const double* A, long long sA, double* B, long long sB, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((M + 31) / 32, (M + 31) / 32, Q), dim3(256), 0, s, A, sA, B, sB, M);
}
void launch_tri_fold(const double* C, double* T, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(tri_fold_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, C, T, M);
}
void launch_kl_terms(const double* Kuui, const double* S, const double* m_u, const double* a, const double* Luu,
                     const double* L, const double* Sqi, int Q, int M, double* out, hipStream_t s) {
  hipLaunchKernelGGL(kl_terms_kernel, dim3(KL_BLOCKS, Q), dim3(256), 0, s, Kuui, S, m_u, a, Luu, L, Sqi, Q, M, out);
}
void launch_dkmm(const double* G, const double* GSK, const double* Kuui, const double* KSK, const double* Kr, const double* a,
                 double* out, int Q, int M, hipStream_t s) {
  if (M >= 64)
    hipLaunchKernelGGL(dkmm_tiled_kernel, dim3((M + 31) / 32, (M + 31) / 32, Q), dim3(256), 0, s, G, GSK, Kuui, KSK, Kr, a, out, M);
  else
    hipLaunchKernelGGL(dkmm_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, G, GSK, Kuui, KSK, Kr, a, out, M);
}
void launch_dlds(const double* G, const double* Kuui, const double* Sqi, double* out, long long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(dlds_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, G, Kuui, Sqi,
                     out, n);
}
void launch_pack_gl(const double* T, double* gL, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(pack_gl_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, T, gL, Q, M);
}
void launch_gmu(const double* Kr, const double* a, double* g, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(gmu_kernel, dim3((M + 255) / 256, Q), dim3(256), 0, s, Kr, a, g, Q, M);
}
void launch_kzz_rows(const double* dKmm, const double* Z, int ldz, int P, const double* d_var, const double* d_ell, int Q,
                     int M, double* rowout, hipStream_t s) {
  dim3 grid((M + 3) / 4, Q);
  DISPATCH_P(P, hipLaunchKernelGGL((kzz_rows_kernel<PP>), grid, dim3(256), 0, s, dKmm, Z, ldz, d_var, d_ell, M, rowout));
}
void launch_qf_combine(const double* p, const double* c, long long ldn, long long N, int Q, int Df, const double* W,
                       const double* kappa, const double* var, double* m, double* v, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(qf_combine_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, p, c, ldn, N, Q, Df, W, kappa,
                     var, m, v);
}

// ---- device-resident Adadelta for q(u) (SVI loop; the recurrence of climin.Adadelta as the reference calls it, util.py:327)
// phase 0 (before the gradient):  x -= pend (the second half-step of the previous iteration);  step1 = m * step;  x -= step1
// phase 1 (after it):             gms = d gms + (1-d) g^2;  step2 = sqrt(sms+o)/sqrt(gms+o) * g * rate;  pend = step2;
//                                 step = step1 + step2;  sms = d sms + (1-d) step^2        with g = sign * grad (or 0)
// i.e. x -= step2 is deferred to the start of the next iteration (same operations in the same order on every element), so
// that between iterations x is the point of the last evaluation -- what the reference's model object holds at that moment
// (climin updates its own `wrt`; the model is only written by stochastic_grad, svmogp.py:188).
// Every operation is a separately rounded IEEE operation in the order of hetmogp_amd/util.py:Adadelta (no FMA contraction),
// so the iterates are bit-identical to the host optimiser's.
namespace {
__global__ void adadelta_kernel(double* __restrict__ x, double* __restrict__ gms, double* __restrict__ sms,
                                double* __restrict__ step, double* __restrict__ pend, const double* __restrict__ grad, double sign,
                                long long n, int phase, double rate, double m, double d, double omd, double o) {
#pragma clang fp contract(off)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const double step1 = step[i] * m;
    if (phase == 0) {
      const double xv = x[i] - pend[i];
      x[i] = xv - step1;
      pend[i] = 0.0;
      continue;
    }
    const double g = grad ? sign * grad[i] : 0.0;
    double t1 = g * g;
    t1 = t1 * omd;
    double gm = gms[i] * d;
    gm = gm + t1;
    gms[i] = gm;
    double a = sqrt(sms[i] + o);
    const double b = sqrt(gm + o);
    a = a / b;
    a = a * g;
    a = a * rate;
    pend[i] = a;
    const double st = step1 + a;
    step[i] = st;
    double t2 = st * st;
    t2 = t2 * omd;
    double sm = sms[i] * d;
    sm = sm + t2;
    sms[i] = sm;
  }
}
}  // namespace

namespace {
// dst = [ head (n_hg) | kl (n_kl) | per-latent tails (Q x n_tail, from bundle + NG + q*per_q + oDZ) | rowout (n_row) ]
__global__ void gather_small_kernel(const double* __restrict__ stats, long long n_hg, const double* __restrict__ kl,
                                    long long n_kl, long long per_q, long long oDZ, long long n_tail, int Q,
                                    const double* __restrict__ rowout, long long n_row, double* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_hg) {
    dst[i] = stats[i];
    return;
  }
  long long k = i - n_hg;
  if (k < n_kl) {
    dst[i] = kl[k];
    return;
  }
  k -= n_kl;
  if (k < n_tail * Q) {
    const long long q = k / n_tail, e = k - q * n_tail;
    dst[i] = stats[n_hg + q * per_q + oDZ + e];
    return;
  }
  k -= n_tail * Q;
  if (k < n_row) dst[i] = rowout[k];
}
}  // namespace

void launch_gather_small(const double* stats, long long n_hg, const double* kl, long long n_kl, long long per_q, long long oDZ,
                         long long n_tail, int Q, const double* rowout, long long n_row, double* dst, hipStream_t s) {
  const long long n = n_hg + n_kl + n_tail * Q + n_row;
  hipLaunchKernelGGL(gather_small_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, stats, n_hg, kl, n_kl, per_q, oDZ,
                     n_tail, Q, rowout, n_row, dst);
}

void launch_adadelta(double* x, double* gms, double* sms, double* step, double* pend, const double* grad, double sign, long long n,
                     int phase, double rate, double m, double d, double omd, double o, hipStream_t s) {
  if (n <= 0) return;
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(adadelta_kernel, dim3(blocks), dim3(256), 0, s, x, gms, sms, step, pend, grad, sign, n, phase, rate, m, d, omd, o);
}

