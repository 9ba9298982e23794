// common.h -- shared declarations of the hetmogp HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#define HMOGP_WAVE 64
#ifndef HMOGP_POTRF_NB
#define HMOGP_POTRF_NB 32  // Cholesky / triangular-inverse panel width
#endif

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ---- host-side error plumbing -----------------------------------------------------------------------
struct HipError {
  hipError_t code;
  const char* what;
  const char* file;
  int line;
};
#define HIP_TRY(expr)                                          \
  do {                                                         \
    hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) throw HipError{_e, #expr, __FILE__, __LINE__}; \
  } while (0)

// ---- wave-level reductions (64 lanes, shuffles -- no LDS) ---------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of `v` for blocks of up to 1024 threads; result valid in thread 0. `scratch` >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

// ---- GEMM ------------------------------------------------------------------------------------------
// C[b] (M x N, row-major, ldc) = alpha * sum_k A(i,k) * s[k] * B(k,j) + beta * C[b]
//   a_kmajor == 0 : A(i,k) = A[i*lda + k]     a_kmajor == 1 : A(i,k) = A[k*lda + i]   (A^T stored)
//   b_kmajor == 1 : B(k,j) = B[k*ldb + j]     b_kmajor == 0 : B(k,j) = B[j*ldb + k]   (B^T stored)
// Batched over `nbatch` (strides sA.. in elements); the LAST batch may be ragged (M_last/N_last/K_last; 0 = same).
// A second, outer batch `nouter` (strides oA..) multiplies it (e.g. pairs x latents in the triangular inverse).
// ksplit > 1 : K is cut in `ksplit` contiguous ranges; range s writes alpha*partial (beta ignored) to
//              C + s*sSplit (the caller reduces the slabs).
// lower_only : only tiles with tile_col <= tile_row are computed (M == N).
struct GemmArgs {
  const double* A = nullptr;
  const double* B = nullptr;
  double* C = nullptr;
  const double* kscale = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldb = 0, ldc = 0;
  long long sA = 0, sB = 0, sC = 0, sS = 0;
  double alpha = 1.0, beta = 0.0;
  int nbatch = 1;
  int M_last = 0, N_last = 0, K_last = 0;
  int nouter = 1;                        // outer batch (grid.y), e.g. the latent index q
  long long oA = 0, oB = 0, oC = 0, oS = 0;
  int ksplit = 1;
  long long sSplit = 0;
  int a_kmajor = 0, b_kmajor = 1;
  int lower_only = 0;
  int diag_balance = 1;  // role 2: balance the diagonal tiles' sub-tiles over the physical SIMDs of the CU (gemm_rowpass.hip)
  int role = 0;  // 0 generic, 1 forward contraction, 2 weighted Gram (names the kernel instantiation for profiles)
  // role 1 only -- row statistics fused into the epilogue while the P~ tile is still in registers (replaces a separate
  // pass over K^ and P~):  for every row n of the tile and this block's 128 columns j
  //   p += K^_nj a_j,  c += P~_nj K^_nj,  pt += K^_nj a_j r2_nj,  ct += P~_nj K^_nj r2_nj      (r2 = |x_n - z_j|^2 / l^2)
  // written as partials per 64-column half tile, fs_part[stat][2 * tile_col + half][n]; launch_combine_parts sums them.
  // Batched over the latents (nbatch): fs_part / fs_a / fs_z / fs_ell / win advance by their strides per batch.
  double* fs_part = nullptr;
  const double* fs_a = nullptr;    // [N]      a = Kuu^-1 m                         (+ batch * fs_sA)
  const double* fs_x = nullptr;    // [M][P]   inputs of the rows (shared by the batch)
  const double* fs_z = nullptr;    // inducing inputs of the latent, row stride fs_ldz (+ batch * fs_sZ)
  const double* fs_ell = nullptr;  // [nbatch] lengthscales (device)
  int fs_ldz = 0, fs_P = 1, fs_hyper = 0;
  long long fs_sPart = 0, fs_sA = 0, fs_sZ = 0;
  int store_c = 1;               // 0: do not write P~ at all (no consumer when the Z gradient is not requested)
  int fs_sq = 0;                 // role 1, specialised kernel: the ONE fused statistic is rowsum(C .* C) of the product itself (slot of
                                 // `c`; fs_a / fs_x / fs_z unused) -- strict q(f): rowsum(T .* T) of T = A L_q without storing T
  const double* fs_k = nullptr;  // role 1, specialised kernel: the fused row statistics re-read THEIR K^ tile from here (same leading
                                 // dimension / batch stride as A) instead of from A -- strict q(f): P~ = A D with K^ a, rowsum(P~ .* K^)
  const double* c_src = nullptr; // with c_sub: C = c_src - op(A) op(B) (same leading dimension / batch stride as C; nullptr: C itself)
  int c_sub = 0;                 // role 1, specialised kernel: C -= op(A) op(B)  (set by its launcher for alpha = -1, beta = 1)
  // Triangular operands: op(A) is M x K, op(B) is K x N; the k-loop of tile (i0, j0) is trimmed to the products that
  // can be non-zero.   a_tri = +1: op(A)[i][k] == 0 for k > i (lower)  -> k < i0 + 128;   -1: == 0 for k < i -> k >= i0
  //                    b_tri = +1: op(B)[k][j] == 0 for k < j (lower)  -> k >= j0;        -1: == 0 for k > j -> k < j0 + 128
  // Uses: forward contraction against T = tril(C) + tril(C^T,-1) when only the quadratic forms k^T C k are wanted (b_tri
  // = +1, half the products); S = L L^T, (L L^T)^-1 = Linv^T Linv, dL/dS L and the triangular-inverse merges.
  int a_tri = 0, b_tri = 0;
  // Exact-zero windows (rowpass.hip: launch_windows): K^ = s2 exp(-r2/2) underflows to exactly 0.0 beyond r ~ 38.6
  // lengthscales, so for spatially sorted rows it is banded.  role 1: win[2*ti], win[2*ti+1] = [lo, hi) column range
  // (multiples of 16) outside which every entry of row tile ti is exactly zero -> column tiles outside it are skipped
  // and the K loop runs over the range only.  role 2: win[2*b], win[2*b+1] = [lo, hi) row range outside which column
  // block b is exactly zero -> the K loop of tile (i,j) runs over the intersection.  Skipped terms are products with
  // exact zeros, so results are unchanged.  nullptr = dense.
  const int* win = nullptr;
  long long win_stride = 0;  // ints per batch
};
// p[n] = sum_t part[0][t][n], c <- part[1], pt <- part[2], ct <- part[3]   (pt/ct may be nullptr)
// batched over nb latents: part + b*sPart, outputs + b*ldn
void launch_combine_parts(const double* part, int tiles, long long n, double* p, double* c, double* pt, double* ct,
                          hipStream_t s, int nb = 1, long long sPart = 0, long long ldn = 0);
void launch_gemm_f64(const GemmArgs& g, hipStream_t stream);
// The two row-pass contractions (role 1 / 2) as specialised 8-wave kernels (gemm_rowpass.hip) when the shape allows it,
// else the general kernel.  Returns the number of fused-row-statistics partials per 128-column tile the role-1 kernel
// wrote (what launch_combine_parts has to sum): 4 (specialised) or 2 (general).
bool gemm_rowpass_eligible(const GemmArgs& g);
void launch_gemm_rowpass(const GemmArgs& g, hipStream_t stream);
int launch_gemm_rowpass_or_general(const GemmArgs& g, hipStream_t stream);
bool gemm_rowpass_would_take(const GemmArgs& g);   // will launch_gemm_rowpass_or_general use the specialised kernel for g?
constexpr int GEMM_MAX_FWD_PARTS = 4;
// 64 x 64-tile kernel for the replicated M x M products (gemm_small.hip); launch_gemm_f64 picks it when the 128-tile grid
// would leave the device under-filled (env HMOGP_SMALL_GEMM=0 disables it).
bool gemm_small_eligible(const GemmArgs& g);
void launch_gemm_small(const GemmArgs& g, hipStream_t stream);

// ---- linear algebra on Q x M x M batches (linalg.hip) ---------------------------------------------------
// In-place lower Cholesky of A[q]; info[q] = 0 or the 1-based index of the first non-positive pivot (LAPACK
// dpotrf convention; the matrix content is then undefined). Upper triangle is zeroed. dscr: Q*M*M doubles (out-of-place factor).
void launch_potrf_batched(double* A, int Q, int M, int* d_info, double* dscr, hipStream_t stream, int panel_begin = 0,
                          int panel_end = -1);
// Linv[q] = L[q]^-1 (lower triangular, upper zero). `L` is preserved; tmp: Q*M*M doubles.  linv_is_zero: the caller has
// already zeroed Linv (off the critical path of a latency-bound chain).
void launch_trtri_batched(const double* L, double* Linv, double* tmp, int Q, int M, hipStream_t stream, bool linv_is_zero = false);
// Out[q] = Linv[q]^T Linv[q]  (= (L L^T)^-1), full symmetric.
void launch_ltl_batched(const double* Linv, double* Out, int Q, int M, hipStream_t stream);

// ---- misc small kernels (linalg.hip) -------------------------------------------------------------------
void launch_fill(double* p, long long n, double v, hipStream_t stream);
void launch_unpack_tril(const double* L_flat /*[Mtri,Q]*/, double* L /*[Q,M,M]*/, int Q, int M, hipStream_t s);
void launch_gemv_batched(const double* A, const double* x, double* y, int Q, int M, long long sx, int incx,
                         hipStream_t s);  // y[q] = A[q] x[q];  x[q][i] = x[q*sx + i*incx]
