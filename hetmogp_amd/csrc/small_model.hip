// small_model.hip -- the replicated M x M algebra of the svmogp_inf path for SMALL models (M <= 64), one block per latent GP
// with every matrix in LDS: TWO launches instead of ~30.
//
// BASELINE.json's config 1 (the reference's own runnable size: N_t = 1000, M = 50, Q = 2; notebooks/demo.ipynb uses M = 8) is
// bound by launch count, not by arithmetic: profiles/r04_C1_kernel_stats_before.csv shows 45 kernels per evaluation, twelve of them
// 50 x 50 x 50 products at ~16 us each.  Everything of util.py:181-200 (K_uu, jitchol, K_uu^-1) and svmogp_inf.py:192-195,
// 227-250 (S, K_uu^-1 S K_uu^-1 - K_uu^-1, S^-1, the KL terms) fits one CU's 160 KB of LDS at this size:
//
//   u_small_kernel       K_uu (GPy rounding order) -> + jitter -> Cholesky -> L_uu^-1 -> K_uu^-1 ; a = K_uu^-1 m ;
//                        L = tril(flat) ; S = L L^T ; K^-1 S ; K^-1 S K^-1 ; C ; tril-fold(C) ; S^-1 ; KL partials
//   finish_small_kernel  H mirrored ; G = K^-1 H K^-1 ; K^-1 r ; dL/dS ; dL/dL (packed) ; dL/dm ; G S K^-1 ; dL/dKmm ; its K_zz-weighted
//                        row sums ; the last block to finish gathers every small result into the caller's page-locked host block
//
// Both write the SAME global buffers as the regular path (engine.hip: Kuu, Luu, Kuui, L, S, KiS, KSK, C, Ctri, Sqi, a, klout /
// G, GSK, dLdS, gL, gmu, Kr, dKmm), so posterior_u / predict_f / natgrad / the debug export work unchanged behind them.
// A failed factorisation (GPy's jitter ladder is needed) only sets info[q]: the engine then repeats the evaluation on the
// regular path, which owns the ladder.  Arithmetic: plain FP64 FMAs on 4 x 4 register micro-tiles (a 64^3 product is
// ~4 us on one CU; MFMA tiles would not be faster at one block per latent) -- results agree with the blocked kernels to rounding.
// The sequential chains (Cholesky, triangular inverses) run on single waves with the matrix in registers: sm_potrf_regs /
// sm_trtri_regs; the LDS-resident versions stay for A/B runs (HMOGP_SMALL_REGS=0).  DESIGN.md 11e has the step-by-step timings.
#include <cstdio>
#include "common.h"
#include "post.h"
#include "rbf_device.h"
#include "small_model.h"
#include "rowpass.h"

namespace {

constexpr int SM = HMOGP_SMALL_M, SLD = SM + 2, NT = 256;   // even leading dimension: 16-byte aligned row pairs

// C = op(A) op(B) (all M x M, LDS, C aliases neither operand): thread (t >> 4, t & 15) owns a 4 x 4 micro-tile.
// Rows / columns beyond M hold garbage that is never stored.  KLO / KHI trim the k-range for triangular operands:
//   tri == 0: k in [0, M);  tri == 1: k <= min(i, j) style handled by the caller through `kmax_of_tile`.
template <bool TA, bool TB>
__device__ __forceinline__ void sm_gemm_fma(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int M,
                                        int kbeg_mode = 0) {
  const int t = threadIdx.x, r0 = 4 * (t >> 4), c0 = 4 * (t & 15);
  if (r0 >= M || c0 >= M) return;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  // kbeg_mode 1: op(A)[i][k] == 0 for k < i and op(B)[k][j] == 0 for k < j (upper x lower: L^-T L^-1): start at max(r0, c0)
  // kbeg_mode 2: op(A)[i][k] == 0 for k > i and op(B)[k][j] == 0 for k > j (lower x upper: L L^T): stop at min(r0, c0) + 3
  const int k0 = kbeg_mode == 1 ? max(r0, c0) : 0;
  const int k1 = kbeg_mode == 2 ? min(M, min(r0, c0) + 4) : M;
  for (int k = k0; k < k1; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = TA ? A[k * SLD + r0 + i] : A[(r0 + i) * SLD + k];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = TB ? B[(c0 + j) * SLD + k] : B[k * SLD + c0 + j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (r0 + i < M && c0 + j < M) C[(r0 + i) * SLD + c0 + j] = acc[i][j];
}

// The same product on the matrix cores: four waves, each a 32 x 32 quadrant = 2 x 2 tiles of v_mfma_f64_16x16x4_f64 (A fragment:
// row = lane & 15, k = lane >> 4; B fragment: k = lane >> 4, column = lane & 15; D: column = lane & 15, row = (lane >> 4) + 4 reg --
// gemm_small.hip's conventions).  k beyond M would bring in the buffers' garbage: those fragment entries are zeros.  Triangular
// k-trimming per quadrant (the operands carry explicit zeros there).  Measured at M = 50 (u_small_kernel's phase stamps): 4.4 us per
// product for the FMA micro-tiles (bound by their LDS reads: 8 ds_read_b64 per 16 FMAs), 4.4 us for a plain MFMA loop (the LDS
// latency of every step exposed), 2.9 us with the fragments of the next 16 k read ahead -- one wave per SIMD issues FP64 MFMAs at
// half rate (135 cycles each here), so 512-thread blocks would halve it again.  HMOGP_SM_GEMM_FMA (compile time) restores the
// micro-tiles.
template <bool TA, bool TB>
__device__ __forceinline__ void sm_gemm(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int M,
                                        int kbeg_mode = 0) {
#ifdef HMOGP_SM_GEMM_FMA
  sm_gemm_fma<TA, TB>(A, B, C, M, kbeg_mode);
#else
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 1, wn = w & 1, lr = lane & 15, lk = lane >> 4;
  if (wm * 32 >= M || wn * 32 >= M) return;
  const int kb = kbeg_mode == 1 ? max(wm, wn) * 32 : 0;
  const int ke = kbeg_mode == 2 ? min(M, min(wm, wn) * 32 + 32) : M;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  // chunks of 16 k (4 MFMA steps): the fragments of the next chunk are read from LDS before the 16 MFMAs of the current one
  // (a step's four reads issued right in front of its MFMAs expose the LDS latency 13 times: 4.4 us per product instead of 1.5)
  auto load_chunk = [&](int kc, double (&fa)[4][2], double (&fb)[4][2]) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int k = kc + 4 * st + lk;
      const bool kin = k < ke;
      const int ka = kin ? k : kb;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + lr, col = wn * 32 + i * 16 + lr;
        const double av = TA ? A[ka * SLD + row] : A[row * SLD + ka];
        const double bv = TB ? B[col * SLD + ka] : B[ka * SLD + col];
        fa[st][i] = kin ? av : 0.0;
        fb[st][i] = kin ? bv : 0.0;
      }
    }
  };
  auto mfma_chunk = [&](int kc, const double (&fa)[4][2], const double (&fb)[4][2]) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
      if (kc + 4 * st < ke) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[st][a], fb[st][b], acc[a][b], 0, 0, 0);
      }
  };
  double fa0[4][2], fb0[4][2], fa1[4][2], fb1[4][2];
  if (kb < ke) load_chunk(kb, fa0, fb0);
  for (int kc = kb; kc < ke; kc += 32) {
    const bool more = kc + 16 < ke;
    if (more) load_chunk(kc + 16, fa1, fb1);
    mfma_chunk(kc, fa0, fb0);
    if (more) {
      if (kc + 32 < ke) load_chunk(kc + 32, fa0, fb0);
      mfma_chunk(kc + 16, fa1, fb1);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = wm * 32 + a * 16 + 4 * r + lk;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = wn * 32 + b * 16 + lr;
        if (i < M && j < M) C[i * SLD + j] = acc[a][b][r];
      }
    }
#endif
}

// Block-wide sums of up to three values at once (one pair of barriers instead of one per value); results valid in thread 0.
// `scratch` >= 3 * NT / 64 doubles.
__device__ __forceinline__ void sm_block_sum3(double& a, double& b, double& c, double* scratch);

__device__ __forceinline__ void sm_block_sum3(double& a, double& b, double& c, double* scratch) {
  a = wave_sum(a), b = wave_sum(b), c = wave_sum(c);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                                         // (scratch may still be read from a previous call)
  if (lane == 0) scratch[w] = a, scratch[NT / 64 + w] = b, scratch[2 * (NT / 64) + w] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = b = c = 0.0;
    for (int i = 0; i < NT / 64; ++i) a += scratch[i], b += scratch[NT / 64 + i], c += scratch[2 * (NT / 64) + i];
  }
}

// Element loops over an M x M matrix: a lane per column, NT / 64 rows per step, SM_IT steps (fixed count: fully unrolled, no
// integer divisions).  Global loads go through sm_fill, which issues all of a thread's loads before the first LDS store -- one
// memory latency per matrix instead of one per element (a runtime-count loop is not software-pipelined by the compiler).
constexpr int SM_RS = NT / 64, SM_IT = SM / SM_RS;
__device__ __forceinline__ bool sm_ij(int it, int M, int& i, int& j) {
  j = threadIdx.x & 63;
  i = (threadIdx.x >> 6) + SM_RS * it;
  return i < M && j < M;
}
template <class LD>
__device__ __forceinline__ void sm_fill(double* __restrict__ X, int M, LD&& ld) {
  double v[SM_IT];
#pragma unroll
  for (int it = 0; it < SM_IT; ++it) {
    int i, j;
    v[it] = sm_ij(it, M, i, j) ? ld(i, j) : 0.0;
  }
#pragma unroll
  for (int it = 0; it < SM_IT; ++it) {
    int i, j;
    if (sm_ij(it, M, i, j)) X[i * SLD + j] = v[it];
  }
}
template <class F>
__device__ __forceinline__ void sm_each(int M, F&& f) {
#pragma unroll
  for (int it = 0; it < SM_IT; ++it) {
    int i, j;
    if (sm_ij(it, M, i, j)) f(i, j);
  }
}
__device__ __forceinline__ void sm_load(double* __restrict__ X, const double* __restrict__ g, int M, long long ldg) {
  sm_fill(X, M, [&](int i, int j) { return g[(long long)i * ldg + j]; });
}
__device__ __forceinline__ void sm_store(const double* __restrict__ X, double* __restrict__ g, int M) {
  sm_each(M, [&](int i, int j) { g[(long long)i * M + j] = X[i * SLD + j]; });
}

// Lower Cholesky of X (in place, lower triangle; the strict upper triangle is left as it was), by ONE wave, left-looking, one
// row per lane.  A wave's LDS operations are processed in order, so the lanes exchange columns through LDS without block
// barriers.  The diagonal stays un-normalised in X until the end (no column product reads it); `diag` receives the pivots.
// Returns LAPACK's info (0, or 1 + the first column whose pivot is <= 0 or NaN); uniform across the wave.
__device__ __forceinline__ int sm_potrf_wave(double* X, double* diag, int M, int lane, int* progress = nullptr) {
  int info = 0;
  for (int j = 0; j < M; ++j) {
    if (lane >= j && lane < M) {   // four interleaved partial sums: the loop is bound by the LDS round trip, not by the FMAs
      const double* xi = X + lane * SLD;
      const double* xj = X + j * SLD;
      double s0 = xi[j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = 0;
      for (; k + 4 <= j; k += 4) {
        s0 = fma(-xi[k], xj[k], s0);
        s1 = fma(-xi[k + 1], xj[k + 1], s1);
        s2 = fma(-xi[k + 2], xj[k + 2], s2);
        s3 = fma(-xi[k + 3], xj[k + 3], s3);
      }
      for (; k < j; ++k) s0 = fma(-xi[k], xj[k], s0);
      X[lane * SLD + j] = (s0 + s1) + (s2 + s3);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double d = X[j * SLD + j];
    if (!(d > 0.0)) {
      info = j + 1;
      break;
    }
    const double sq = sqrt(d);
    __builtin_amdgcn_wave_barrier();
    if (lane == j) diag[j] = sq;
    else if (lane > j && lane < M) X[lane * SLD + j] = X[lane * SLD + j] / sq;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // column j (hence row j of L, pivot in diag[j]) is final: a wave that follows may consume it (LDS operations of one wave are
    // processed in order: the counter lands behind the column)
    if (progress && lane == 0) __hip_atomic_store(progress, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (info && progress && lane == 0) __hip_atomic_store(progress, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (!info && lane < M) X[lane * SLD + lane] = diag[lane];
  return info;
}

// Xi = L^-1 (L lower triangular in LDS), zeros above the diagonal: lane c owns column c (forward substitution of L x = e_c;
// it only ever reads its own column of Xi back).  One wave; the other waves of the block may do independent work meanwhile.
// The same, one row BEHIND a Cholesky factorisation that another wave of the block is still running: row r of L is final once the
// factorisation has finished column r (`progress` > r; -1 = it failed); the pivots are read from `diag` (the factorisation keeps
// the diagonal of X un-normalised until its end).  The two sequential chains overlap: ~27 + 23 us -> ~30 us at M = 50.
__device__ __forceinline__ void sm_trtri_follow(const double* L, const double* diag, double* Xi, int M, int lane, int* progress) {
  for (int r = 0; r < M; ++r) {
    for (;;) {
      const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (p < 0) return;
      if (p > r) break;
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    const double* lr = L + r * SLD;
    double s0 = (lane == r) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= r; k += 4) {
      s0 = fma(-lr[k], Xi[k * SLD + lane], s0);
      s1 = fma(-lr[k + 1], Xi[(k + 1) * SLD + lane], s1);
      s2 = fma(-lr[k + 2], Xi[(k + 2) * SLD + lane], s2);
      s3 = fma(-lr[k + 3], Xi[(k + 3) * SLD + lane], s3);
    }
    for (; k < r; ++k) s0 = fma(-lr[k], Xi[k * SLD + lane], s0);
    const double v = (lane <= r) ? ((s0 + s1) + (s2 + s3)) / diag[r] : 0.0;
    if (lane < SM) Xi[r * SLD + lane] = v;
  }
}

__device__ __forceinline__ void sm_trtri_wave(const double* __restrict__ L, double* __restrict__ Xi, int M, int lane) {
  // row by row: x[r][c] = (delta_rc - sum_{k<r} L[r][k] x[k][c]) / L[r][r]; rows above the diagonal are zeros, so every lane runs
  // the same k-range (uniform loop, L[r][k] is an LDS broadcast, x[k][c] a conflict-free row access)
  for (int r = 0; r < M; ++r) {
    const double* lr = L + r * SLD;
    double s0 = (lane == r) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= r; k += 4) {
      s0 = fma(-lr[k], Xi[k * SLD + lane], s0);
      s1 = fma(-lr[k + 1], Xi[(k + 1) * SLD + lane], s1);
      s2 = fma(-lr[k + 2], Xi[(k + 2) * SLD + lane], s2);
      s3 = fma(-lr[k + 3], Xi[(k + 3) * SLD + lane], s3);
    }
    for (; k < r; ++k) s0 = fma(-lr[k], Xi[k * SLD + lane], s0);
    const double v = (lane <= r) ? ((s0 + s1) + (s2 + s3)) / lr[r] : 0.0;
    if (lane < SM) Xi[r * SLD + lane] = v;
  }
}

// ---- the same three wave-level chains with the wave's matrix in REGISTERS --------------------------------------------------------
// The LDS versions above pay an LDS round trip (~100+ cycles) for every dependent step: ~1100 cycles per column / row at M = 50
// (27 us for the factorisation, 25 us for a triangular inverse).  Here lane i keeps row i of the factorised matrix (lane c:
// column c of the inverse) in 64 registers, the loops are fully unrolled (register indices are compile-time constants), and LDS
// only carries what crosses lanes: the finished column (one write, broadcast reads) / the finished rows of L.
typedef __attribute__((address_space(3))) const double sm_lds_cd;   // LDS pointer that stays one (ds_read, not flat_load)
__device__ __forceinline__ double sm_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sm_wave_sync() {   // LDS operations of one wave are processed in order: only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Lower Cholesky factor of X (M x M in LDS; in place: lower triangle + diagonal, zeros above), right-looking, by one wave.
// `col` is a 64-double LDS scratch.  diag[] receives the pivots, `progress` counts finished columns (-1: failed) for a follower.
// Returns LAPACK's info.
__device__ __forceinline__ void sm_sqrt_pair(double d, double& sq, double& rs) {
  // sqrt(d) and 1 / sqrt(d) from the hardware estimate and two coupled (Goldschmidt) refinements -- ~10 dependent operations
  // instead of the ~45 of an IEEE sqrt followed by an IEEE division.  Both results are within 1 ulp; pivots of a covariance
  // matrix are nowhere near the ends of the exponent range.
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g), h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g), h = fma(h, e, h);
  sq = g, rs = h + h;
}
// `col`: 2 x 64 doubles of LDS scratch (the finished column, double-buffered); rdiag[] receives the reciprocal pivots.
__device__ __forceinline__ int sm_potrf_regs(double* X, double* diag, double* rdiag, double* col, int M, int lane, int* progress) {
  double a[SM];
#pragma unroll
  for (int k = 0; k < SM; ++k) a[k] = (lane < M && k < M) ? X[lane * SLD + k] : 0.0;
  sm_lds_cd* colv = (sm_lds_cd*)col;                     // (an opaque base register: LDS offsets beyond 64 KB do not fit the
  asm volatile("" : "+v"(colv));                         //  instruction's offset field and would be materialised per read)
  int info = 0;
  double d = sm_readlane(a[0], 0), sq, rs;
  if (!(d > 0.0)) info = 1;
  sm_sqrt_pair(d, sq, rs);
#pragma unroll
  for (int j = 0; j < SM; ++j) {
    if (j >= M || info) continue;                        // (uniform; no early exit: the loop must unroll completely)
    const double l = (lane > j) ? a[j] * rs : 0.0;       // column j below the diagonal (zero in the lanes above it)
    a[j] = (lane == j) ? sq : l;
    X[lane * SLD + j] = a[j];                            // column j of L (explicit zeros above the diagonal)
    col[(j & 1) * SM + lane] = l;
    if (lane == j) diag[j] = sq, rdiag[j] = rs;
    sm_wave_sync();
    if (progress && lane == 0) __hip_atomic_store(progress, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sm_lds_cd* cj = colv + (j & 1) * SM;
    double c[SM];
#pragma unroll
    for (int k = j + 1; k < SM; ++k) c[k] = cj[k];       // broadcast reads of the column: all in flight ...
    __builtin_amdgcn_sched_barrier(0);                   // ... before the first FMA (the scheduler would serialise them otherwise)
    // look-ahead: the next column's own element first, then its pivot and the refinement chain -- independent of the rest of the
    // trailing update below, which hides their latency (one wave, in-order issue: the compiler interleaves the two)
    if (j + 1 < SM) {
      a[j + 1] = fma(-l, c[j + 1], a[j + 1]);
      d = sm_readlane(a[j + 1], j + 1);
      sm_sqrt_pair(d, sq, rs);
    }
#pragma unroll
    for (int k = j + 2; k < SM; ++k) a[k] = fma(-l, c[k], a[k]);   // trailing update
    if (j + 1 < M && !(d > 0.0)) info = j + 2;
  }
  if (info && progress && lane == 0) __hip_atomic_store(progress, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return info;
}
// spins until the factorisation has finished column r; false: it failed
__device__ __noinline__ bool sm_wait_progress(int* progress, int r) {
  for (;;) {
    const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (p < 0) return false;
    if (p > r) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
  return true;
}
// `rdiag`: the reciprocals of L's diagonal in LDS (the factorisation's by-product, or sm_rdiag).
template <bool FOLLOW>
__device__ __forceinline__ void sm_trtri_regs(const double* L, const double* rdiag, double* Xi, int M, int lane, int* progress) {
  double x[SM];
  bool dead = false;
  sm_lds_cd* Lv = (sm_lds_cd*)L;                         // (opaque base register: see sm_potrf_regs)
  asm volatile("" : "+v"(Lv));
#pragma unroll
  for (int r = 0; r < SM; ++r) {
    if (r >= M || dead) continue;                        // (uniform; no early exit: the loop must unroll completely)
    if (FOLLOW) {
      dead = !sm_wait_progress(progress, r);
      if (dead) continue;
    }
    sm_lds_cd* lr = Lv + r * SLD;
    double s0 = (lane == r) ? 1.0 : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) {
      const double t = lr[k];
      if (k >= (r & ~3) || (k & 3) == 0) s0 = fma(-t, x[k], s0);
      else if ((k & 3) == 1) s1 = fma(-t, x[k], s1);
      else if ((k & 3) == 2) s2 = fma(-t, x[k], s2);
      else s3 = fma(-t, x[k], s3);
    }
    if (FOLLOW) {     // (the row of L only exists now: all of its reads in flight before the first FMA)
      __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 80, 0);
    }
    const double v = (lane <= r) ? ((s0 + s1) + (s2 + s3)) * rdiag[r] : 0.0;
    x[r] = v;
    Xi[r * SLD + lane] = v;
  }
}

// grid (Q, 2): block (q, 0) runs the K_uu chain and then the joint part, block (q, 1) the q(u) chain -- the two sequential
// wave-level chains (Cholesky + triangular inverse of K_uu; triangular inverse of L) run on two CUs at once.  Block (q, 0) needs
// S from block (q, 1): handed over through HBM behind an agent-scope release / acquire on flag[q] (2 Q <= 16 blocks: always
// co-resident; S is published first thing, so the wait is never taken in practice).
#define SM_STAMP(i)                                                                                   \
  do {                                                                                                \
    if (u.stamps && threadIdx.x == 0) u.stamps[((long long)blockIdx.x * 2 + blockIdx.y) * 16 + (i)] = (long long)__builtin_readcyclecounter(); \
  } while (0)

template <int P>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void u_small_kernel(SmallU u) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* X0 = lds;
  double* X1 = X0 + SM * SLD;
  double* X2 = X1 + SM * SLD;
  double* X3 = X2 + SM * SLD;
  double* vec = X3 + SM * SLD;          // [6][SM]: diag pivots | m | a | scratch (2) | reciprocal pivots
  __shared__ double red[16];
  __shared__ int s_info, s_progress;
  const int q = blockIdx.x, role = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int M = u.M, Q = u.Q;
  const long long MM = (long long)M * M, off = (long long)q * MM;
  double* o = u.klout + (long long)q * KL_BLOCKS * 5;

  const int seq = (int)u.seq[0];
  SM_STAMP(0);
  if (role == 1) {
    // ---- q(u) chain: L = flat_to_triang(L_flat) (svmogp_inf.py:193), S = L L^T (:194-195), S^-1 = dpotri(L) (:124) ------------
    for (long long e = (long long)q * NT + t; e < u.nzero; e += (long long)Q * NT) u.zero[e] = 0.0;   // (the bundle: see SmallU)
    sm_fill(X2, M, [&](int r, int c) { return (c <= r) ? u.Lflat[((long long)r * (r + 1) / 2 + c) * Q + q] : 0.0; });
    __syncthreads();
    SM_STAMP(1);
    sm_store(X2, u.L + off, M);
    sm_gemm<false, true>(X2, X2, X0, M, 2);                // S = L L^T  -> X0
    __syncthreads();
    SM_STAMP(2);
    sm_store(X0, u.S + off, M);
    __threadfence();                                       // S is in HBM (agent scope) ...
    __syncthreads();
    if (t == 0) __hip_atomic_store(&u.flag[q], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... before block (q, 0) may read it
    SM_STAMP(3);
    double l2 = 0.0, ninf = 0.0;
    if (t < M) l2 = log(fabs(X2[t * SLD + t]));
    if (w == 0) {                                          // L^-1 -> X3 (one wave)
      if (u.regs) {
        if (lane < M) vec[5 * SM + lane] = 1.0 / X2[lane * SLD + lane];   // reciprocal pivots: one division for all rows
        sm_wave_sync();
        sm_trtri_regs<false>(X2, vec + 5 * SM, X3, M, lane, nullptr);
      } else
        sm_trtri_wave(X2, X3, M, lane);
    }
    __syncthreads();
    SM_STAMP(4);
    sm_gemm<true, false>(X3, X3, X1, M, 1);                // S^-1 = L^-T L^-1  -> X1
    __syncthreads();
    SM_STAMP(5);
    sm_store(X1, u.Sqi + off, M);
    sm_each(M, [&](int i, int j) { ninf += isinf(X1[i * SLD + j]) ? 1.0 : 0.0; });
    double unused = 0.0;
    sm_block_sum3(l2, ninf, unused, red);
    if (t == 0) o[5 + 0] = 0.0, o[5 + 1] = 0.0, o[5 + 2] = 0.0, o[5 + 3] = l2, o[5 + 4] = ninf;   // KL partial "block 1"
    SM_STAMP(6);
    return;
  }

  // ---- K_uu chain ------------------------------------------------------------------------------------------------------------
  const double var = u.var[q], ell = u.ell[q], jit = u.jit[q];
  // K_uu = k_q(Z_q, Z_q), both arguments passed (util.py:197): GPy's rounding order, no forced diagonal
  {
    double* zs = X3;                                     // (free until the joint part) inducing inputs of the latent: [M][P]
    for (int e = t; e < M * P; e += NT) zs[e] = u.Z[(long long)(e / P) * u.ldz + q * P + (e % P)];
    if (t < M) vec[SM + t] = u.mu[(long long)t * Q + q];
    __syncthreads();
    sm_each(M, [&](int i, int j) {
      double zi[P], zj[P];
#pragma unroll
      for (int p = 0; p < P; ++p) zi[p] = zs[i * P + p], zj[p] = zs[j * P + p];
      const double r2 = rbf_r2<P>(zi, sumsq<P>(zi), zj, sumsq<P>(zj), ell);
      const double k = var * exp(-0.5 * r2);
      u.Kuu[off + (long long)i * M + j] = k;
      X0[i * SLD + j] = k + ((i == j) ? jit : 0.0);     // the factorised copy carries the jitter (GPy jitchol)
    });
  }
  if (t == 0) s_info = 0, s_progress = 0;
  for (int e = t; e < KL_BLOCKS * 5; e += NT)          // KL partials: blocks 0 and 1 are written below / by block (q, 1)
    if (e >= 10) o[e] = 0.0;
  for (int e = t; e < KL_BLOCKS; e += NT)              // ... and the block maxima of diag(K_uu^-1) behind them: slot 0 below
    if (e >= 1) u.klout[(long long)Q * KL_BLOCKS * 5 + (long long)q * KL_BLOCKS + e] = 0.0;
  __syncthreads();
  SM_STAMP(1);
  if (u.stop_after == 1) return;
  if (u.stop_after == 6) return;
  if (w == 0) {                                        // wave 0: L_uu = chol(K_uu + jitter I) ...
    const int info = u.regs ? sm_potrf_regs(X0, vec, vec + 5 * SM, vec + 3 * SM, M, lane, &s_progress)
                            : sm_potrf_wave(X0, vec, M, lane, &s_progress);
    if (lane == 0) s_info = info;
  } else if (w == 1 && u.stop_after != 5) {            // ... wave 1, one row behind it: L_uu^-1 -> X1
    if (u.regs) sm_trtri_regs<true>(X0, vec + 5 * SM, X1, M, lane, &s_progress);
    else sm_trtri_follow(X0, vec, X1, M, lane, &s_progress);
  } else if (w == 2) {                                 // ... wave 2 meanwhile: S of block (q, 1) (published first thing) -> X3
    if (lane == 0)
      while (__hip_atomic_load(&u.flag[q], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double* Sg = u.S + off;
    for (int i0 = 0; i0 < M; i0 += 8) {
      double v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (i0 + e < M && lane < M) ? Sg[(long long)(i0 + e) * M + lane] : 0.0;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (i0 + e < M && lane < M) X3[(i0 + e) * SLD + lane] = v[e];
    }
  }
  __syncthreads();
  SM_STAMP(2);
  if (t == 0) u.info[q] = s_info;                        // (non-zero: the engine falls back to the regular path and its ladder)
  if (u.stop_after == 2 || u.stop_after == 5) return;
  if (s_info) return;
  // L_uu with an explicit zero upper triangle (potrf_finalize)
  sm_each(M, [&](int i, int j) { u.Luu[off + (long long)i * M + j] = (j <= i) ? X0[i * SLD + j] : 0.0; });
  double l1 = 0.0;
  if (t < M) l1 = log(fabs(X0[t * SLD + t]));
  __syncthreads();
  sm_gemm<true, false>(X1, X1, X0, M, 1);                // K_uu^-1 = L_uu^-T L_uu^-1            (util.py:199)            -> X0
  __syncthreads();
  SM_STAMP(3);
  sm_store(X0, u.Kuui + off, M);
  double ma = 0.0, tr = 0.0;
  if (t < M) {                                           // a = K_uu^-1 m
    double sacc = 0.0;
    for (int k = 0; k < M; ++k) sacc = fma(X0[t * SLD + k], vec[SM + k], sacc);
    u.a[(long long)q * M + t] = sacc;
    ma = vec[SM + t] * sacc;
  }
  SM_STAMP(4);
  if (u.stop_after == 3) return;
  // ---- joint part: needs S of block (q, 1) -- in X3 since the factorisation (wave 2) -------------------------------------------
  SM_STAMP(5);
  sm_each(M, [&](int i, int j) { tr += X0[i * SLD + j] * X3[i * SLD + j]; });
  sm_gemm<false, false>(X0, X3, X1, M);                  // K^-1 S                                                         -> X1
  __syncthreads();
  SM_STAMP(6);
  sm_store(X1, u.KiS + off, M);
  sm_gemm<false, false>(X1, X0, X2, M);                  // K^-1 S K^-1                                                    -> X2
  __syncthreads();
  SM_STAMP(7);
  sm_store(X2, u.KSK + off, M);
  sm_each(M, [&](int i, int j) {                         // C = K^-1 S K^-1 - K^-1 ; T = tril(C) + tril(C^T, -1)
    const double cij = X2[i * SLD + j] - X0[i * SLD + j];
    u.C[off + (long long)i * M + j] = cij;
    double tv = 0.0;
    if (j == i) tv = cij;
    else if (j < i) tv = cij + (X2[j * SLD + i] - X0[j * SLD + i]);
    u.Ctri[off + (long long)i * M + j] = tv;
  });
  // KL partials (svmogp_inf.py:245-249), kl_terms_kernel's layout: "block 0" of the latent
  double kmax = (t < M) ? X0[t * SLD + t] : 0.0;       // max_i (K_uu^-1)_ii (kl_terms_kernel's condition estimate; M <= 64: wave 0)
  for (int o2 = 32; o2; o2 >>= 1) kmax = fmax(kmax, __shfl_xor(kmax, o2, 64));
  sm_block_sum3(tr, ma, l1, red);
  if (t == 0) o[0] = tr, o[1] = ma, o[2] = l1, o[3] = 0.0, o[4] = 0.0;
  if (t == 0) u.klout[(long long)Q * KL_BLOCKS * 5 + (long long)q * KL_BLOCKS] = kmax;
  SM_STAMP(8);
}

// rowout[q][m] = { sum_j EK_mj, sum_j EK_mj r2_mj, sum_j (EK_mj + EK_jm)(z_j - z_m)[p] }, EK = dKmm .* K_zz: kzz_rows_kernel's
// arithmetic (one wave per row, lanes stride over j, wave_sum), dKmm read from LDS
template <int P>
__device__ __forceinline__ void sm_kzz_rows(const SmallF& f, const double* __restrict__ D, int q) {
  const int M = f.M, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* Zq = f.Z + (long long)q * P;
  const double v = f.var[q], l = f.ell[q];
  for (int m = w; m < M; m += NT / 64) {
    double zm[P];
#pragma unroll
    for (int p = 0; p < P; ++p) zm[p] = Zq[(long long)m * f.ldz + p];
    const double zmsq = sumsq<P>(zm);
    double s1 = 0.0, s2 = 0.0, gz[P];
#pragma unroll
    for (int p = 0; p < P; ++p) gz[p] = 0.0;
    for (int j = lane; j < M; j += 64) {
      double zj[P];
#pragma unroll
      for (int p = 0; p < P; ++p) zj[p] = Zq[(long long)j * f.ldz + p];
      double r2 = rbf_r2<P>(zm, zmsq, zj, sumsq<P>(zj), l);
      if (j == m) r2 = 0.0;
      const double kz = v * exp(-0.5 * r2);
      const double ek = D[m * SLD + j] * kz, ekt = D[j * SLD + m] * kz;
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
      double* o = f.rowout + ((long long)q * M + m) * (2 + P);
      o[0] = s1;
      o[1] = s2;
#pragma unroll
      for (int p = 0; p < P; ++p) o[2 + p] = gz[p];
    }
  }
}

__device__ __forceinline__ void finish_small_body(const SmallF& f, double* lds) {
  double* X0 = lds;
  double* X1 = X0 + SM * SLD;
  double* X2 = X1 + SM * SLD;
  double* X3 = X2 + SM * SLD;
  double* vec = X3 + SM * SLD;          // [4][SM]: r | K^-1 r | a
  // grid (Q, 2): both blocks of a latent form G = K^-1 H K^-1 and K^-1 r (two products: cheaper than a hand-over); block (q, 0)
  // goes on with the q(u) gradients, block (q, 1) with the K_uu-side ones -- the two tails run on two CUs at once
  const int q = blockIdx.x, role = blockIdx.y, t = threadIdx.x;
  const int M = f.M, Q = f.Q;
  const long long MM = (long long)M * M, off = (long long)q * MM;
  const double* Hq = f.H + (long long)q * f.per_q;
  if (role == 1 && !f.want_hz) return;
  // H_q arrives as its lower triangle (row pass / exchange step): mirrored here
  sm_fill(X0, M, [&](int i, int j) { return (j <= i) ? Hq[(long long)i * M + j] : Hq[(long long)j * M + i]; });
  sm_load(X1, f.Kuui + off, M, M);
  if (t < M) vec[t] = Hq[f.oR + t], vec[2 * SM + t] = f.a[(long long)q * M + t];
  __syncthreads();
  if (role == 0) {
    // (the bundle itself keeps the full symmetric H_q, like launch_mirror_lower)
    sm_each(M, [&](int i, int j) {
      if (j > i) f.Hfull[(long long)q * f.per_q + (long long)i * M + j] = X0[i * SLD + j];
    });
  }
  sm_gemm<false, false>(X0, X1, X2, M);                  // H K^-1                                                X2
  if (t < M) {                                           // K^-1 r  (dVE_dmu, svmogp_inf.py:144)
    double s = 0.0;
    for (int k = 0; k < M; ++k) s = fma(X1[t * SLD + k], vec[k], s);
    vec[SM + t] = s;
    if (role == 0) f.Kr[(long long)q * M + t] = s;
    if (role == 0 && f.want_qu) {                                        // dL/dm = K^-1 r - a   (:130,144,168)
      f.gmu[(long long)t * Q + q] = s - vec[2 * SM + t];
      if (f.gmu2) f.gmu2[(long long)t * Q + q] = s - vec[2 * SM + t];
    }
  }
  __syncthreads();
  sm_gemm<false, false>(X1, X2, X3, M);                  // G = K^-1 (H K^-1)  (dVE_dS, svmogp_inf.py:148)          X3
  __syncthreads();
  // The regular path forms the lower tiles of G = K^-1 H K^-1 and mirrors them: exactly symmetric.  Same here.
  sm_each(M, [&](int i, int j) {
    if (j > i) X3[i * SLD + j] = X3[j * SLD + i];
  });
  __syncthreads();
  if (role == 0) sm_store(X3, f.G + off, M);
  if (role == 0 && f.want_qu) {
    sm_load(X0, f.Sqi + off, M, M);                      // (H is no longer needed)
    __syncthreads();
    sm_each(M, [&](int i, int j) {                       // dL/dS = G - (K^-1 - S^-1) / 2   (svmogp_inf.py:131,169)
      const double v = X3[i * SLD + j] - 0.5 * (X1[i * SLD + j] - X0[i * SLD + j]);
      X2[i * SLD + j] = v;
      f.dLdS[off + (long long)i * M + j] = v;
    });
    __syncthreads();
    sm_load(X0, f.L + off, M, M);
    __syncthreads();
    sm_gemm<false, false>(X2, X0, X1, M);                // dL/dS L (:175-177)  [X1: K^-1 is re-read from HBM below]
    __syncthreads();
    sm_each(M, [&](int r, int c) {                       // GPy triang_to_flat of 2 dL/dS L
      if (c <= r) {
        const long long o = ((long long)r * (r + 1) / 2 + c) * Q + q;
        f.gL[o] = 2.0 * X1[r * SLD + c];
        if (f.gL2) f.gL2[o] = 2.0 * X1[r * SLD + c];
      }
    });
    __syncthreads();
  }
  if (role == 1) {
    sm_load(X0, f.KiS + off, M, M);
    __syncthreads();
    sm_gemm<false, true>(X3, X0, X2, M);                 // G S K^-1 = G (K^-1 S)^T   (tmp_dv, svmogp_inf.py:151)    X2
    __syncthreads();
    sm_store(X2, f.GSK + off, M);
    // dL_dKmm (svmogp_inf.py:130-133,151-154,166,170): dkmm_kernel's formula
    sm_load(X0, f.KSK + off, M, M);                      // (K^-1 S is no longer needed; K^-1 is still in X1)
    __syncthreads();
    sm_each(M, [&](int i, int j) {
      const double kri = vec[SM + i], krj = vec[SM + j], ai = vec[2 * SM + i], aj = vec[2 * SM + j];
      const double xij = X3[i * SLD + j] - X2[i * SLD + j] - X2[j * SLD + i] - kri * aj;
      const double xji = X3[j * SLD + i] - X2[j * SLD + i] - X2[i * SLD + j] - krj * ai;
      const double dve = 0.5 * (xij + xji);
      const double dkl = 0.5 * X1[i * SLD + j] - 0.5 * X0[i * SLD + j] - 0.5 * (ai * aj);
      f.dKmm[off + (long long)i * M + j] = dve - dkl;
      X0[i * SLD + j] = dve - dkl;
    });
    if (f.rowout) {
      __syncthreads();
      switch (f.P) {
        case 1: sm_kzz_rows<1>(f, X0, q); break;
        case 2: sm_kzz_rows<2>(f, X0, q); break;
        case 3: sm_kzz_rows<3>(f, X0, q); break;
        default: sm_kzz_rows<4>(f, X0, q); break;
      }
    }
  }
}

__global__ __launch_bounds__(NT) void finish_small_kernel(SmallF f) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int s_last;
  finish_small_body(f, lds);
  if (!f.stage) return;
  // ---- the last block to get here gathers the results into the host block -------------------------------------------------------
  __threadfence();
  __syncthreads();
  const int t = threadIdx.x;
  if (t == 0) s_last = atomicAdd(f.counter, 1) == (int)(gridDim.x * gridDim.y) - 1;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // (other blocks' results: not from a stale L1 line)
  // (plain copy loops, unrolled: a thread's loads of a segment are in flight together -- one branchy loop over the whole block
  //  pays a memory latency per element)
  auto copy = [&](double* __restrict__ dst, const double* __restrict__ src, long long n) {
#pragma unroll 8
    for (long long i = t; i < n; i += NT) dst[i] = src[i];
  };
  double* d = f.stage;
  copy(d, f.g_stats, f.n_hg), d += f.n_hg;
  copy(d, f.g_kl, f.n_kl), d += f.n_kl;
  for (int q = 0; q < f.Q; ++q) copy(d, f.g_stats + f.NG + q * f.per_q + f.oDZ, f.n_tail), d += f.n_tail;
  copy(d, f.rowout, f.n_row), d += f.n_row;
  copy(d, f.g_extra, f.n_extra), d += f.n_extra;
  if (t < f.Q) d[t] = (double)f.g_info[t];
  if (t == 0) *f.counter = 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// [r4] The ROW PASS of a small model as two kernels (64 rows per block, latents one after the other, K^ / C_q tiles in LDS):
//   small_fwd_kernel  K^ = k_q(X, Z_q) (the hot path's rounding variant, rbf_kernel<P, false>) -> HBM; P~ = K^ C_q on 4 x 4 FMA
//                     micro-tiles; row statistics p, c (and the r2-weighted p~, c~) reduced over the 16 lanes that share a row
//                     -- replaces rbf + forward GEMM + combine_parts (svmogp_inf.py:212-218; util.py:145-164)
//   small_bwd_kernel  block partials of H_q = K^T diag(beta) K^ (lower micro-tiles), r_q = K^T alpha, dZ_q (svmogp_inf.py:145-147,
//                     157-161 reduced against K^) -- replaces colstats + weighted Gram + two slab reductions
//   small_red_kernel  deterministic sum of the block partials into the statistic bundle
constexpr int RB = 64;

template <int P>
__global__ __launch_bounds__(NT) void small_fwd_kernel(SmallRows a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Cs = lds;                     // [M][SLD]
  double* Kt = Cs + SM * SLD;           // [RB][SLD]
  double* av = Kt + RB * SLD;           // [SM]
  double* zs = av + SM;                 // [SM][P] inducing inputs / lengthscale
  double* xs = zs + SM * P;             // [RB][P] inputs / lengthscale
  const int t = threadIdx.x, M = a.M;
  const long long n0 = (long long)blockIdx.x * RB;
  const int nr = (int)min((long long)RB, a.n - n0);
  const int r0 = 4 * (t >> 4), c0 = 4 * (t & 15);
  {                                      // one latent per block: grid (row blocks, Q)
    const int q = blockIdx.y;
    const double var = a.var[q], ell = a.ell[q], il2 = 1.0 / (ell * ell), inv_l = 1.0 / ell;
    const double* Cq = a.C + (long long)q * M * M;
    sm_load(Cs, Cq, M, M);
    if (t < M) av[t] = a.a[(long long)q * M + t];
    for (int e = t; e < M * P; e += NT) zs[e] = a.Z[(long long)(e / P) * a.ldz + q * P + (e % P)];
    for (int e = t; e < RB * P; e += NT) xs[e] = (e / P < nr) ? a.X[(n0 + e / P) * P + (e % P)] : 0.0;
    __syncthreads();
    double* Khq = a.Kh + (long long)q * a.ldn * M;
#pragma unroll
    for (int it = 0; it < SM_IT; ++it) {        // K^ tile: rbf_kernel<P, false>'s arithmetic (clip(r2) / l^2, no sqrt / divide)
      const int m = t & 63, r = (t >> 6) + SM_RS * it;
      if (m < M) {
        double k = 0.0;
        if (r < nr) {
          double xv[P], zv[P];
#pragma unroll
          for (int p = 0; p < P; ++p) xv[p] = xs[r * P + p], zv[p] = zs[m * P + p];
          const double r2 = rbf_r2_fast<P>(xv, sumsq<P>(xv), zv, sumsq<P>(zv), il2);
          k = var * exp(-0.5 * r2);
          Khq[(n0 + r) * M + m] = k;
        }
        Kt[r * SLD + m] = k;
      }
    }
    __syncthreads();
    // P~ micro-tile: rows r0..r0+3 of the block, columns c0..c0+3
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    if (c0 < M)
      for (int k = 0; k < M; ++k) {
        double ka[4], cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ka[i] = Kt[(r0 + i) * SLD + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) cb[j] = Cs[k * SLD + c0 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fma(ka[i], cb[j], acc[i][j]);
      }
    double sp[4], sc[4], spt[4], sct[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sp[i] = sc[i] = spt[i] = sct[i] = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = c0 + j;
        if (m < M) {
          const double kv = Kt[(r0 + i) * SLD + m], pv = acc[i][j];
          sp[i] += kv * av[m];
          sc[i] += pv * kv;
          if (a.hyper) {          // r2 as the regular epilogue forms it: sum_p (x/l - z/l)^2
            double r2 = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const double d = xs[(r0 + i) * P + p] * inv_l - zs[m * P + p] * inv_l;
              r2 += d * d;
            }
            const double wv = kv * r2;
            spt[i] += wv * av[m];
            sct[i] += pv * wv;
          }
        }
      }
    }
#pragma unroll
    for (int o = 1; o <= 8; o <<= 1)             // the 16 lanes (t & 15) that share rows r0..r0+3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sp[i] += __shfl_xor(sp[i], o, 64);
        sc[i] += __shfl_xor(sc[i], o, 64);
        if (a.hyper) spt[i] += __shfl_xor(spt[i], o, 64), sct[i] += __shfl_xor(sct[i], o, 64);
      }
    if ((t & 15) == 0)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (r0 + i < nr) {
          const long long o = (long long)q * a.ldn + n0 + r0 + i;
          a.vp[o] = sp[i], a.vc[o] = sc[i];
          if (a.hyper) a.vpt[o] = spt[i], a.vct[o] = sct[i];
        }
    if (a.want_z && c0 < M) {
      double* Ptq = a.Pt + (long long)q * a.ldn * M;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (r0 + i < nr)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < M) Ptq[(n0 + r0 + i) * M + c0 + j] = acc[i][j];
    }
    __syncthreads();
  }
}

template <int P>
__global__ __launch_bounds__(NT) void small_bwd_kernel(SmallRows a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Kt = lds;                     // [RB][SLD]  K^ rows of the block
  double* Pl = Kt + RB * SLD;           // [RB][SLD]  P~ rows (Z gradient only)
  double* av = Pl + RB * SLD;           // [SM]
  double* zs = av + SM;                 // [SM][P]
  double* xs = zs + SM * P;             // [RB][P]
  double* wt = xs + RB * P;             // [4][RB] alpha | beta | alpha0 | beta0
  const int t = threadIdx.x, M = a.M;
  const long long n0 = (long long)blockIdx.x * RB;
  const int nr = (int)min((long long)RB, a.n - n0);
  const long long slab_q = (long long)M * M + M + (long long)M * P;
  const int m0 = 4 * (t >> 4), c0 = 4 * (t & 15);
  {                                      // one latent per block: grid (row blocks, Q)
    const int q = blockIdx.y;
    const double* Khq = a.Kh + (long long)q * a.ldn * M;
    const double* Ptq = a.Pt + (long long)q * a.ldn * M;
    {                                       // K^ (and P~) rows of the block: all loads of a thread in flight at once
      double kv[SM_IT], pv[SM_IT];
#pragma unroll
      for (int it = 0; it < SM_IT; ++it) {
        const int m = t & 63, r = (t >> 6) + SM_RS * it;
        const bool in = m < M && r < nr;
        kv[it] = in ? Khq[(n0 + r) * M + m] : 0.0;
        pv[it] = (in && a.want_z) ? Ptq[(n0 + r) * M + m] : 0.0;
      }
#pragma unroll
      for (int it = 0; it < SM_IT; ++it) {
        const int m = t & 63, r = (t >> 6) + SM_RS * it;
        if (m < M) {
          Kt[r * SLD + m] = kv[it];
          if (a.want_z) Pl[r * SLD + m] = pv[it];
        }
      }
    }
    if (t < M) av[t] = a.a[(long long)q * M + t];
    for (int e = t; e < M * P; e += NT) zs[e] = a.Z[(long long)(e / P) * a.ldz + q * P + (e % P)];
    for (int e = t; e < RB * P; e += NT) xs[e] = (e / P < nr) ? a.X[(n0 + e / P) * P + (e % P)] : 0.0;
    if (t < RB) {
      const long long o = (long long)q * a.ldn + n0 + t;
      const bool in = t < nr;
      wt[t] = in ? a.alpha[o] : 0.0, wt[RB + t] = in ? a.beta[o] : 0.0;
      wt[2 * RB + t] = in ? a.alpha0[o] : 0.0, wt[3 * RB + t] = in ? a.beta0[o] : 0.0;
    }
    __syncthreads();
    double* out = a.slab + ((long long)blockIdx.x * a.Q + q) * slab_q;
    // H_q partial: lower micro-tiles only (the bundle carries the lower triangle between begin and finish)
    if (m0 < M && c0 < M && c0 <= m0 + 3) {
      double acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
      for (int r = 0; r < nr; ++r) {
        const double b = wt[RB + r];
        double ka[4], kb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ka[i] = Kt[r * SLD + m0 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) kb[j] = Kt[r * SLD + c0 + j] * b;     // k-scaled B operand, like the MFMA Gram
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fma(ka[i], kb[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (m0 + i < M && c0 + j <= m0 + i) out[(long long)(m0 + i) * M + c0 + j] = acc[i][j];
    }
    // r_q = K^T alpha ; dZ_q[m] = sum_n E_nm (x_n - z_m), E = (alpha0 a^T + 2 diag(beta0) P~) .* K^
    if (t < M) {
      const int m = t;
      double rs = 0.0, dz[P];
#pragma unroll
      for (int p = 0; p < P; ++p) dz[p] = 0.0;
      const double am = av[m];
      for (int r = 0; r < nr; ++r) {
        const double k = Kt[r * SLD + m];
        rs += k * wt[r];
        if (a.want_z) {
          const double e = (wt[2 * RB + r] * am + 2.0 * wt[3 * RB + r] * Pl[r * SLD + m]) * k;
#pragma unroll
          for (int p = 0; p < P; ++p) dz[p] += e * (xs[r * P + p] - zs[m * P + p]);
        }
      }
      out[(long long)M * M + m] = rs;
#pragma unroll
      for (int p = 0; p < P; ++p) out[(long long)M * M + M + (long long)m * P + p] = dz[p];
    }
    __syncthreads();
  }
}

// bundle += sum over the blocks' partials, block by block in order (deterministic); one thread per output element
__global__ __launch_bounds__(NT) void small_red_kernel(SmallRows a, int nblk, SmallQuadRed qr) {
  if (blockIdx.y == a.Q) {
    // the quadrature's scalars: one wave per SLOT (over the whole plane of blocks), the segments one after the other -- two tasks
    // add to the same word of the bundle only through the same slot (sum ve, #(v<0), sa_q, sl_q), so the order is fixed
    const int lane = threadIdx.x & 63, gw = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nw = gridDim.x * (NT / 64);
    for (int k = gw; k < HMOGP_MAXSCAL; k += nw)
      for (int sg = 0; sg < qr.nseg; ++sg) {
        const auto& g = qr.s[sg];
        if (k >= g.nscal) continue;
        double s = 0.0;
        for (long long b = lane; b < g.nrows; b += 64) s += g.part[b * g.nscal + k];
        s = wave_sum(s);
        if (lane == 0) a.stats[g.off[k]] += s;
      }
    return;
  }
  const int q = blockIdx.y, M = a.M, P = a.P;
  const long long slab_q = (long long)M * M + M + (long long)M * P;
  const long long e = (long long)blockIdx.x * NT + threadIdx.x;
  if (e >= slab_q) return;
  long long dst;
  if (e < (long long)M * M) {
    const int i = (int)(e / M), j = (int)(e - (long long)i * M);
    if (j > i) return;                                   // (upper triangle: never written by the blocks)
    dst = a.NG + q * a.per_q + e;
  } else if (e < (long long)M * M + M) {
    dst = a.NG + q * a.per_q + a.oR + (e - (long long)M * M);
  } else {
    if (!a.want_z) return;
    dst = a.NG + q * a.per_q + a.oDZ + (e - (long long)M * M - M);
  }
  // eight interleaved partial sums in a FIXED order (blocks b = i mod 8): eight independent loads in flight, same bits every run
  double sacc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const double* src = a.slab + (long long)q * slab_q + e;
  const long long st = (long long)a.Q * slab_q;
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(b + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) sacc[u] += v[u];
  }
  for (; b < nblk; ++b) sacc[0] += src[(long long)b * st];
  a.stats[dst] += ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + ((sacc[4] + sacc[5]) + (sacc[6] + sacc[7]));
}
}  // namespace

size_t small_lds_bytes() { return sizeof(double) * (4 * SM * SLD + 6 * SM); }

void launch_u_small(const SmallU& u_in, hipStream_t s) {
  SmallU u = u_in;
  static const int stop = [] {
    const char* e = getenv("HMOGP_USMALL_STOP");
    return e ? atoi(e) : 0;
  }();
  u.stop_after = stop;
  static const int regs = [] {   // HMOGP_SMALL_REGS=0: the LDS-resident factorisation / inverses (A/B runs)
    const char* e = getenv("HMOGP_SMALL_REGS");
    return e ? atoi(e) : 1;
  }();
  u.regs = regs;
  // HMOGP_USMALL_STAMPS=1: shader-clock stamps at the phase boundaries of every block, printed every 1000 launches
  static long long* stamps = [] {
    long long* p = nullptr;
    const char* e = getenv("HMOGP_USMALL_STAMPS");
    if (e && atoi(e)) (void)hipHostMalloc((void**)&p, sizeof(long long) * 8 * 2 * 16, hipHostMallocDefault);
    return p;
  }();
  static int n_launch = 0;
  if (stamps) {
    if (++n_launch % 1000 == 0) {
      (void)hipStreamSynchronize(s);
      for (int q = 0; q < u.Q; ++q)
        for (int r = 0; r < 2; ++r) {
          const long long* t = stamps + (q * 2 + r) * 16;
          std::fprintf(stderr, "u_small block (%d, %d): start %+lld vs (0,0);", q, r, t[0] - stamps[0]);
          for (int i = 1; i <= (r ? 6 : 8); ++i) std::fprintf(stderr, " %d:%lld", i, t[i] - t[0]);
          std::fprintf(stderr, "\n");
        }
    }
    u.stamps = stamps;
  }
  static bool attr_set = false;
  if (!attr_set) {   // > 64 KB of dynamic LDS needs the opt-in
    HIP_TRY(hipFuncSetAttribute((const void*)u_small_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes()));
    HIP_TRY(hipFuncSetAttribute((const void*)u_small_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes()));
    HIP_TRY(hipFuncSetAttribute((const void*)u_small_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes()));
    HIP_TRY(hipFuncSetAttribute((const void*)u_small_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes()));
    HIP_TRY(hipFuncSetAttribute((const void*)finish_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes()));
    attr_set = true;
  }
  DISPATCH_P(u.P, hipLaunchKernelGGL((u_small_kernel<PP>), dim3(u.Q, 2), dim3(NT), small_lds_bytes(), s, u));
}

void launch_finish_small(const SmallF& f, hipStream_t s) {
  hipLaunchKernelGGL(finish_small_kernel, dim3(f.Q, 2), dim3(NT), small_lds_bytes(), s, f);
}

size_t small_rows_lds_bytes() { return sizeof(double) * (2 * SM * SLD + SM + SM * 4 + RB * 4 + 4 * RB); }

static void small_rows_attr() {
  static bool done = false;
  if (done) return;
  const int b = (int)small_rows_lds_bytes();
  HIP_TRY(hipFuncSetAttribute((const void*)small_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  HIP_TRY(hipFuncSetAttribute((const void*)small_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, b));
  done = true;
}

void launch_small_fwd(const SmallRows& r, hipStream_t s) {
  if (r.n <= 0) return;
  small_rows_attr();
  const unsigned nblk = (unsigned)((r.n + 63) / 64);
  DISPATCH_P(r.P, hipLaunchKernelGGL((small_fwd_kernel<PP>), dim3(nblk, r.Q), dim3(NT), small_rows_lds_bytes(), s, r));
}

void launch_small_bwd(const SmallRows& r, hipStream_t s, const SmallQuadRed* qr) {
  if (r.n <= 0) return;
  small_rows_attr();
  const unsigned nblk = (unsigned)((r.n + 63) / 64);
  DISPATCH_P(r.P, hipLaunchKernelGGL((small_bwd_kernel<PP>), dim3(nblk, r.Q), dim3(NT), small_rows_lds_bytes(), s, r));
  const long long slab_q = (long long)r.M * r.M + r.M + (long long)r.M * r.P;
  SmallQuadRed none;
  hipLaunchKernelGGL(small_red_kernel, dim3((unsigned)((slab_q + NT - 1) / NT), r.Q + (qr && qr->nseg ? 1 : 0)), dim3(NT), 0, s, r,
                     (int)nblk, qr ? *qr : none);
}
