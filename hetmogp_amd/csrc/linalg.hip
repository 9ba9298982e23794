// linalg.hip -- batched M x M linear algebra of the replicated part of the svmogp_inf path (gfx950).
//
//   potrf  : blocked right-looking lower Cholesky, LDS-resident 32 x 32 diagonal panels, trailing update on the
//            FP64-MFMA GEMM.  Replaces LAPACK dpotrf behind GPy jitchol (hetmogp/util.py:198).
//   trtri  : triangular inverse by LDS diagonal-block inverses + log2(M/32) levels of batched GEMM merges.
//   ltl    : (L L^T)^-1 = Linv^T Linv.  trtri + ltl replace LAPACK dpotri behind GPy dpotri
//            (hetmogp/util.py:199, hetmogp/svmogp_inf.py:124).
#include "common.h"

namespace {

constexpr int NB = HMOGP_POTRF_NB;  // panel width (common.h)
constexpr int NBP = NB + 1;  // padded LDS leading dimension

// ---------------------------------------------------------------------------------------------- potrf
// One launch per panel j.  Every block factorises the diagonal block redundantly in LDS (32^3/3 flops) and solves
// 256 rows of the panel below it (one row per thread, L_jj read as LDS broadcasts).  The factorised diagonal block
// is NOT written in place (late blocks of this launch still read the unfactorised one): block 0 parks it in
// `dscr` ([Q][M][NB]) and potrf_finalize_kernel scatters it at the end.  info[q] != 0 makes the latent a no-op.
__global__ __launch_bounds__(256) void potrf_panel_kernel(double* __restrict__ Aall, int M, int j, int* __restrict__ info,
                                                          double* __restrict__ dscr) {
  __shared__ double D[NB][NBP];
  __shared__ int fail;
  const int q = blockIdx.y;
  if (info[q] != 0) return;
  double* A = Aall + (long long)q * M * M;
  const int jb = min(NB, M - j);
  const int t = threadIdx.x;
  if (t == 0) fail = 0;
  for (int e = t; e < NB * NB; e += blockDim.x) {
    const int r = e / NB, c = e % NB;
    D[r][c] = (r < jb && c <= r) ? A[(long long)(j + r) * M + (j + c)] : 0.0;
  }
  __syncthreads();
  // The diagonal block is factorised by the first wave entirely in registers: lane r owns row r (32 doubles); column c
  // needs the pivot from lane c and, for the rank-1 update, the scaled column entries L[c2][c] from lanes c2 -- both by
  // wavefront shuffles (about 530 of them for the whole block), no LDS round trips or barriers inside the recurrence.
  if (t < 64) {
    const int r = t & (NB - 1);  // NB = 32: lanes 32..63 mirror lanes 0..31 (shuffles stay within the wave)
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = D[r][c];
    int bad = 0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const double d = __shfl(a[c], c, 64);
      if (c < jb && !bad && !(d > 0.0)) bad = c + 1;  // LAPACK: ajj <= 0 or NaN (uniform across lanes)
      const double piv = sqrt(d);
      const double l = (r == c) ? piv : ((r > c) ? a[c] / piv : a[c]);
      a[c] = l;
#pragma unroll
      for (int c2 = c + 1; c2 < NB; ++c2) {
        const double lc2 = __shfl(l, c2, 64);
        if (r >= c2) a[c2] -= l * lc2;
      }
    }
    if (t < jb)
#pragma unroll
      for (int c = 0; c < NB; ++c) D[r][c] = a[c];
    if (t == 0) fail = bad;
  }
  __syncthreads();
  if (fail) {
    if (blockIdx.x == 0 && t == 0) info[q] = j + fail;
    return;
  }
  if (blockIdx.x == 0) {
    double* ds = dscr + (long long)q * M * NB;
    for (int e = t; e < jb * NB; e += blockDim.x) {
      const int r = e / NB, c = e % NB;
      ds[(long long)(j + r) * NB + c] = (c <= r) ? D[r][c] : 0.0;
    }
  }
  // panel rows below the diagonal block:  x L_jj^T = a
  const int row = j + jb + blockIdx.x * blockDim.x + t;
  if (row < M) {
    double* a = A + (long long)row * M + j;
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = (c < jb) ? a[c] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      if (c < jb) {
        double s = x[c];
#pragma unroll
        for (int k = 0; k < NB; ++k)
          if (k < c) s -= x[k] * D[c][k];
        x[c] = s / D[c][c];
      }
    }
#pragma unroll
    for (int c = 0; c < NB; ++c)
      if (c < jb) a[c] = x[c];
  }
}

// Zero the strict upper triangle and scatter the parked diagonal blocks.
__global__ void potrf_finalize_kernel(double* __restrict__ A, int M, const double* __restrict__ dscr) {
  const int q = blockIdx.z;
  double* a = A + (long long)q * M * M;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= M) return;
  if (c > r)
    a[(long long)r * M + c] = 0.0;
  else if (c / NB == r / NB)
    a[(long long)r * M + c] = dscr[((long long)q * M + r) * NB + (c % NB)];
}

// ---------------------------------------------------------------------------------------------- trtri
// Inverse of each 32 x 32 lower-triangular diagonal block (thread c solves L x = e_c).
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double* __restrict__ Lall, double* __restrict__ Xall, int M) {
  __shared__ double D[NB][NBP];
  __shared__ double X[NB][NBP];
  const int q = blockIdx.y, j = blockIdx.x * NB, jb = min(NB, M - j), t = threadIdx.x;
  const double* L = Lall + (long long)q * M * M;
  double* Xo = Xall + (long long)q * M * M;
  for (int e = t; e < NB * NB; e += blockDim.x) {
    const int r = e / NB, c = e % NB;
    D[r][c] = (r < jb && c <= r) ? L[(long long)(j + r) * M + (j + c)] : 0.0;
    X[r][c] = 0.0;
  }
  __syncthreads();
  if (t < jb) {
    const int c = t;
    X[c][c] = 1.0 / D[c][c];
    for (int r = c + 1; r < jb; ++r) {
      double s = 0.0;
      for (int k = c; k < r; ++k) s += D[r][k] * X[k][c];
      X[r][c] = -s / D[r][r];
    }
  }
  __syncthreads();
  for (int e = t; e < jb * jb; e += blockDim.x) {
    const int r = e / jb, c = e % jb;
    Xo[(long long)(j + r) * M + (j + c)] = X[r][c];
  }
}

// ---------------------------------------------------------------------------------------------- misc
__global__ void fill_kernel(double* __restrict__ p, long long n, double v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// L[q][r][c] = L_flat[(r(r+1)/2 + c) * Q + q] for c <= r (GPy choleskies.flat_to_triang), else 0.
__global__ void unpack_tril_kernel(const double* __restrict__ Lf, double* __restrict__ L, int Q, int M) {
  const int q = blockIdx.z, r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= M) return;
  const long long idx = (long long)r * (r + 1) / 2 + c;
  L[((long long)q * M + r) * M + c] = (c <= r) ? Lf[idx * Q + q] : 0.0;
}

// y[q][i] = sum_j A[q][i][j] x[q][j]; one wave per row.
__global__ __launch_bounds__(256) void gemv_kernel(const double* __restrict__ A, const double* __restrict__ x,
                                                   double* __restrict__ y, int M, long long sx, int incx) {
  const int q = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const double* a = A + ((long long)q * M + row) * M;
  const double* xv = x + (long long)q * sx;
  double s = 0.0;
  for (int j = lane; j < M; j += 64) s += a[j] * xv[(long long)j * incx];
  s = wave_sum(s);
  if (lane == 0) y[(long long)q * M + row] = s;
}

}  // namespace

void launch_fill(double* p, long long n, double v, hipStream_t s) {
  if (n <= 0) return;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
}

void launch_unpack_tril(const double* L_flat, double* L, int Q, int M, hipStream_t s) {
  hipLaunchKernelGGL(unpack_tril_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, s, L_flat, L, Q, M);
}

void launch_gemv_batched(const double* A, const double* x, double* y, int Q, int M, long long sx, int incx,
                         hipStream_t s) {
  hipLaunchKernelGGL(gemv_kernel, dim3((M + 3) / 4, Q), dim3(256), 0, s, A, x, y, M, sx, incx);
}

void launch_potrf_batched(double* A, int Q, int M, int* d_info, double* dscr, hipStream_t stream) {
  HIP_TRY(hipMemsetAsync(d_info, 0, sizeof(int) * Q, stream));
  for (int j = 0; j < M; j += NB) {
    const int jb = std::min(NB, M - j);
    const int rem = M - j - jb;
    const int nblk = std::max(1, (rem + 255) / 256);
    hipLaunchKernelGGL(potrf_panel_kernel, dim3(nblk, Q), dim3(256), 0, stream, A, M, j, d_info, dscr);
    if (rem > 0) {
      GemmArgs g;
      g.A = A + (long long)(j + jb) * M + j;
      g.B = g.A;
      g.C = A + (long long)(j + jb) * M + (j + jb);
      g.M = g.N = rem;
      g.K = jb;
      g.lda = g.ldb = g.ldc = M;
      g.nbatch = Q;
      g.sA = g.sB = g.sC = (long long)M * M;
      g.alpha = -1.0;
      g.beta = 1.0;
      g.a_kmajor = 0;
      g.b_kmajor = 0;
      g.lower_only = 1;
      launch_gemm_f64(g, stream);
    }
  }
  hipLaunchKernelGGL(potrf_finalize_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, stream, A, M, dscr);
}

// Linv = L^-1.  `tmp` is a Q x M x M scratch.
void launch_trtri_batched(const double* L, double* Linv, double* tmp, int Q, int M, hipStream_t stream) {
  const long long MM = (long long)M * M;
  HIP_TRY(hipMemsetAsync(Linv, 0, sizeof(double) * MM * Q, stream));
  hipLaunchKernelGGL(trtri_diag_kernel, dim3((M + NB - 1) / NB, Q), dim3(64), 0, stream, L, Linv, M);
  // merge [[X11, 0], [X21, X22]] with X21 = -X22 * L21 * X11, block size s doubling
  for (int s = NB; s < M; s *= 2) {
    const int npairs = (M - s - 1) / (2 * s) + 1;  // pairs whose right block is non-empty
    const int last_right = (2 * (npairs - 1) + 1) * s;
    const int r_last = std::min(s, M - last_right);
    const long long pstride = (long long)2 * s * M + 2 * s;
    // T = L21 * X11      (r x s) = (r x s)(s x s)
    GemmArgs g;
    g.A = L + (long long)s * M;
    g.lda = M;
    g.a_kmajor = 0;
    g.B = Linv;
    g.ldb = M;
    g.b_kmajor = 1;
    g.C = tmp + (long long)s * M;
    g.ldc = M;
    g.M = s;
    g.N = s;
    g.K = s;
    g.M_last = r_last;
    g.nbatch = npairs;
    g.sA = g.sB = g.sC = pstride;
    g.nouter = Q;
    g.oA = g.oB = g.oC = MM;
    launch_gemm_f64(g, stream);
    // X21 = -X22 * T     (r x s) = (r x r)(r x s)
    GemmArgs h;
    h.A = Linv + (long long)s * M + s;
    h.lda = M;
    h.a_kmajor = 0;
    h.B = tmp + (long long)s * M;
    h.ldb = M;
    h.b_kmajor = 1;
    h.C = Linv + (long long)s * M;
    h.ldc = M;
    h.M = s;
    h.N = s;
    h.K = s;
    h.M_last = r_last;
    h.K_last = r_last;
    h.nbatch = npairs;
    h.sA = h.sB = h.sC = pstride;
    h.nouter = Q;
    h.oA = h.oB = h.oC = MM;
    h.alpha = -1.0;
    launch_gemm_f64(h, stream);
  }
}

void launch_ltl_batched(const double* Linv, double* Out, int Q, int M, hipStream_t stream) {
  GemmArgs g;
  g.A = Linv;
  g.B = Linv;
  g.C = Out;
  g.M = g.N = g.K = M;
  g.lda = g.ldb = g.ldc = M;
  g.a_kmajor = 1;
  g.b_kmajor = 1;
  g.nbatch = Q;
  g.sA = g.sB = g.sC = (long long)M * M;
  launch_gemm_f64(g, stream);
}
