// linalg.hip -- batched M x M linear algebra of the replicated part of the svmogp_inf path (gfx950).
//
//   potrf  : blocked right-looking lower Cholesky, one fused launch per 32-column panel (diagonal block in registers,
//            panel solve, FP64-MFMA trailing update).  Replaces LAPACK dpotrf behind GPy jitchol (hetmogp/util.py:198).
//   trtri  : triangular inverse by in-register 64 x 64 diagonal-block inverses + log2(M/64) levels of batched GEMM merges.
//   ltl    : (L L^T)^-1 = Linv^T Linv.  trtri + ltl replace LAPACK dpotri behind GPy dpotri
//            (hetmogp/util.py:199, hetmogp/svmogp_inf.py:124).
#include "common.h"

long long* g_potrf_stamps = nullptr;  // probe_potrf.hip (-DPOTRF_STAMPS): phase time stamps of the first step launch

namespace {

#ifdef POTRF_STAMPS
#define STAMP(i) do { if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

constexpr int NB = HMOGP_POTRF_NB;  // panel width (common.h)
constexpr int NBP = NB + 1;  // padded LDS leading dimension

// ---------------------------------------------------------------------------------------------- potrf
// Blocked right-looking lower Cholesky, ONE launch per 32-column panel j (the chain is latency-bound: 32 dependent
// launches instead of the 64 of a separate panel-solve + trailing-GEMM pair).  Block (ti, tj), ti >= tj, owns the
// 128 x 128 tile (ti, tj) of the trailing matrix W[j+32.., j+32..] and does everything that tile needs itself:
//   1. factorises the 32 x 32 diagonal block W[j.., j..] (redundantly per block, first wave, in registers);
//   2. solves the panel rows of row tiles ti and tj against it (one row per thread: threads 0..127 -> tile ti,
//      128..255 -> tile tj) and parks them k-major in LDS;
//   3. W_tile -= L21_i L21_j^T on the FP64 matrix cores (v_mfma_f64_16x16x4_f64, 4 waves x 64 x 64).
// The factor is written OUT of place (`Lo`): blocks of this launch still read the
// unfactorised panel of W.  Blocks (ti, 0) write the solved rows of tile ti, block 0 the diagonal block.
// info[q] != 0 (LAPACK's pivot test: d <= 0 or NaN at column info-1) makes the latent a no-op for the later launches.
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // wave-uniform broadcast through SGPRs
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Value of half `HC` of the wave (lanes 32 HC .. 32 HC + 31) in BOTH halves: v_permlane32_swap_b32 of a register with itself
// returns {lower half in both halves, upper half in both halves}.
template <int HC>
__device__ __forceinline__ double bcast_half(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[HC], (int)a[HC]);
}

// One column of the in-wave factorisation (C is a compile-time constant: every register index and lane number below is).
template <int C>
__device__ __forceinline__ void factor_diag_column(double (&a)[NB / 2], double* Dinv, double (*Lv)[NB], int jb, int lane, int r,
                                                   int h, int& bad) {
  constexpr int HC = C & 1, KC = C >> 1;
  const double d = readlane_f64(a[KC], C + 32 * HC);
  if (C < jb && !bad && !(d > 0.0)) bad = C + 1;  // LAPACK: ajj <= 0 or NaN (uniform across lanes)
  // pivot sqrt(d) and its reciprocal from v_rsq_f64 + Newton steps (a dependent chain of ~12 FMAs instead of an IEEE sqrt
  // followed by an IEEE division; both end within an ulp or two of the correctly rounded values)
  double y = __builtin_amdgcn_rsq(d);
  y = y * fma(-0.5 * d * y, y, 1.5);
  y = y * fma(-0.5 * d * y, y, 1.5);
  double piv = d * y;
  piv = fma(0.5 * y, fma(-piv, piv, d), piv);
  const double rinv = fma(y, fma(-piv, y, 1.0), y);
  if (lane == C) Dinv[C] = rinv;
  const double lown = (r == C) ? piv : a[KC] * rinv;  // column C of L, meaningful in the half that owns the column
  if (h == HC) a[KC] = lown;
  if (C + 1 < NB) {
    const double l = bcast_half<HC>(lown);            // l[r] = L[r][C] in both halves
    // the next pivot column first, straight from registers (it is the critical path) ...
    constexpr int HN = (C + 1) & 1, KN = (C + 1) >> 1;
    const double sn = readlane_f64(l, C + 1);
    a[KN] = fma(-((h == HN) ? l : 0.0), sn, a[KN]);
    // ... the other columns with the l vector broadcast through LDS: lane (r, h) needs l[2k + h] for its columns
    Lv[C & 1][r] = l;
#pragma unroll
    for (int k = KC + 1; k < NB / 2; ++k) {
      const double sk = Lv[C & 1][2 * k + h];
      // C odd: column 2 (KC + 1) of half 0 is the next pivot column, already updated above
      const double lk = (HC == 1 && k == KC + 1) ? ((h == 1) ? l : 0.0) : l;
      a[k] = fma(-lk, sk, a[k]);
    }
  }
}

template <int C>
__device__ __forceinline__ void factor_diag_columns(double (&a)[NB / 2], double* Dinv, double (*Lv)[NB], int jb, int lane, int r,
                                                    int h, int& bad) {
  if constexpr (C < NB) {
    factor_diag_column<C>(a, Dinv, Lv, jb, lane, r, h, bad);
    factor_diag_columns<C + 1>(a, Dinv, Lv, jb, lane, r, h, bad);
  }
}

// Factorisation of the 32 x 32 diagonal block by ONE wave, in registers: lane (r, h) = r + 32 h owns the 16 columns 2k + h
// of row r, so both halves of the wave carry half of every rank-1 update.  Column c needs the pivot from the lane that
// owns (c, c) (v_readlane, compile-time lane); the scaled column l = L[:, c] is mirrored into the other half with
// v_permlane32_swap; the NEXT pivot column is updated at once from a v_readlane broadcast (the recurrence's critical path),
// all other columns from the l vector parked in LDS (per-lane addresses 2k + h: a uniform SGPR broadcast cannot serve two
// halves that need different entries).  Every element sees the same operations in the same order as the plain
// right-looking recurrence: a[c2] = fma(-L[r][c], L[c2][c], a[c2]) for c = 0, 1, ...  (3.4x fewer instructions than one
// row per lane with 31 - c broadcasts per column.)  Entries above the diagonal carry garbage that nothing reads.
// Returns LAPACK's info (0, or 1 + the failing column).
__device__ __forceinline__ int factor_diag_wave(double (*D)[NBP + 1], double* Dinv, double (*Lv)[NB], int jb, int lane) {
  static_assert(NB == 32, "one row per lane of a half wave");
  const int r = lane & (NB - 1), h = lane >> 5;
  double a[NB / 2];
#pragma unroll
  for (int k = 0; k < NB / 2; ++k) a[k] = D[r][2 * k + h];
  int bad = 0;
  factor_diag_columns<0>(a, Dinv, Lv, jb, lane, r, h, bad);
#pragma unroll
  for (int k = 0; k < NB / 2; ++k) D[r][2 * k + h] = a[k];
  return bad;
}

// Where the reciprocal pivot c of the diagonal block at column j is parked in the out-of-place factor between the launch that
// factorises the block ahead of time and the launch that consumes it: strictly above the diagonal of that block (nothing
// else ever touches the upper triangle of `Lo`): (c, c + 1) for c <= 30, (0, 2) for c = 31.
__device__ __forceinline__ long long dinv_slot(int j, int c, int M) {
  return (c < NB - 1) ? (long long)(j + c) * M + (j + c + 1) : (long long)j * M + (j + 2);
}

// [r4] TS = edge of a trailing-matrix tile.  rocprofv3 (profiles/r04_C3_kernel_stats.csv) shows the chain is bound by the
// DURATION of a step kernel (23.9 us at M = 1024, 41 us at M = 2048), not by the launch boundary (~1.5 us): the largest piece
// is the tile update -- with 128 x 128 tiles one wave per SIMD issues 128 MFMAs at half rate (~6.8 us).  TS = 64 gives four
// times as many blocks (408 at M = 1024, Q = 3) with 32 MFMAs per wave.
template <int TS>
__global__ __launch_bounds__(320) void potrf_step_kernel(double* __restrict__ Wall, double* __restrict__ Lall, int M, int j,
                                                         int* __restrict__ info, long long* stamps, int pre) {
  STAMP(0);
  constexpr int WT = TS / 2, NS = WT / 16;      // wave tile (2 x 2 waves per block tile), 16 x 16 sub-tiles per wave-tile edge
  constexpr int LSP = TS + 16;                  // k-major leading dimension of the parked panel rows (bank argument: gemm_f64.hip)
  __shared__ __attribute__((aligned(16))) double D[NB][NBP + 1];  // even leading dimension: 16-byte column pairs
  __shared__ __attribute__((aligned(16))) double Ls[2][NB][LSP];
  __shared__ double Dinv[NB];  // reciprocals of the pivots
  __shared__ __attribute__((aligned(16))) double Lv[2][NB];  // factor wave: the current column of L, for both half waves
  __shared__ int fail;
  const int q = blockIdx.y;
  if (info[q] != 0) return;
  double* W = Wall + (long long)q * M * M;
  double* Lo = Lall + (long long)q * M * M;
  const int jb = min(NB, M - j), base = j + jb, rem = M - base;
  const int t = threadIdx.x;
  // The LAST block of a launch with rows below the panel is the look-ahead block: it repeats the panel solve of tile 0 and
  // then only factorises the next diagonal block (below) -- on a CU of its own, because FP64 vector work shares the pipe
  // with the FP64 MFMAs of a tile update (inside block 0 the same factorisation took twice as long).
  const bool ahead_blk = rem > 0 && blockIdx.x == gridDim.x - 1;
  int ti = 0, tj = 0;
  if (!ahead_blk) {
    const int v = blockIdx.x;
    ti = (int)((sqrt(8.0 * (double)v + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= v) ++ti;
    while (ti * (ti + 1) / 2 > v) --ti;
    tj = v - ti * (ti + 1) / 2;
  }
  if (t == 0) fail = 0;
  // pre != 0: the previous launch already factorised this diagonal block (look-ahead, below) into its final place in Lo
  {
    const double* src = pre ? Lo : W;
    for (int e = t; e < NB * NB; e += blockDim.x) {
      const int r = e / NB, c = e % NB;
      D[r][c] = (r < jb && c <= r) ? src[(long long)(j + r) * M + (j + c)] : 0.0;
    }
    if (pre && rem > 0 && t < NB) Dinv[t] = Lo[dinv_slot(j, t, M)];
  }
  __syncthreads();
  STAMP(1);
  // Wave 4 (threads 256..319) factorises the diagonal block while the others have their panel rows in flight.
  const int half = (t / TS) & 1, tl = t % TS;
  const int prow = base + (half ? tj : ti) * TS + tl;
  const bool pvalid = t < 2 * TS && rem > 0 && prow < M;  // (a ragged last panel, jb < NB, has no rows below it: rem == 0)
  double x[NB];
  if (t >= 256) {
    if (!pre) {
      const int bad = factor_diag_wave(D, Dinv, Lv, jb, t & 63);
      if (t == 256) fail = bad;
    }
  } else {
    // Panel rows of the two row tiles (threads 0..TS-1 -> tile ti, TS..2TS-1 -> tile tj): in flight while wave 4 factorises.
    if (pvalid) {
      const double* w = W + (long long)prow * M + j;
      if ((M & 1) == 0) {
#pragma unroll
        for (int c = 0; c < NB; c += 2) {
          const f64x2 v2 = *reinterpret_cast<const f64x2*>(w + c);
          x[c] = v2.x, x[c + 1] = v2.y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NB; ++c) x[c] = w[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < NB; ++c) x[c] = 0.0;
    }
  }
  __syncthreads();
  STAMP(2);
  if (fail) {
    if (blockIdx.x == 0 && t == 0) info[q] = j + fail;
    return;
  }
  if (blockIdx.x == 0 && !pre)
    for (int e = t; e < jb * NB; e += blockDim.x) {
      const int r = e / NB, c = e % NB;
      if (c <= r) Lo[(long long)(j + r) * M + (j + c)] = D[r][c];
    }
  if (rem <= 0) return;
  // x L_jj^T = w, right-looking: the updates of one column are independent FMAs (L_jj read as uniform LDS broadcasts)
  if (t < 2 * TS) {
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      x[c] *= Dinv[c];
#pragma unroll
      for (int k = c + 1; k < NB; ++k) x[k] = fma(-x[c], D[k][c], x[k]);
    }
    if (pvalid && half == 0 && tj == 0 && !ahead_blk) {
      double* lo = Lo + (long long)prow * M + j;
#pragma unroll
      for (int c = 0; c < NB; ++c) lo[c] = x[c];
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) Ls[half][c][tl] = x[c];
  }
  STAMP(7);
  // This block's tile of the trailing matrix goes straight into the MFMA accumulators (D fragment: col = lane & 15,
  // row = (lane >> 4) + 4 * reg); the loads are in flight across the barrier.
  const int lane = t & 63, wv = t >> 6, wm = wv >> 1, wn = wv & 1, lr = lane & 15, lk = lane >> 4;
  const bool idle = t >= 256 || ahead_blk || (ti == tj && wm == 0 && wn == 1);  // factor wave; look-ahead block; strictly-upper quadrant of a diagonal tile
  f64x4 acc[NS][NS];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = base + ti * TS + wm * WT + a * 16 + 4 * r + lk;
#pragma unroll
      for (int b = 0; b < NS; ++b) {
        const int col = base + tj * TS + wn * WT + b * 16 + lr;
        acc[a][b][r] = (!idle && row < M && col <= row) ? W[(long long)row * M + col] : 0.0;
      }
    }
  // Look-ahead (the factor wave of the look-ahead block): the NEXT diagonal block = W[base.., base..] - X X^T with X the
  // first 32 solved rows, formed with the same MFMA sequence on the same operands as the main update of tile (0, 0)
  // (bit-identical values), factorised while the regular blocks update the trailing matrix, and written to its final
  // place in Lo: the next launch starts its panel solve at once instead of waiting for a factor wave of its own.
  const bool ahead = ahead_blk && t >= 256;
  const int jbn = min(NB, rem);
  f64x4 la[2][2];
  if (ahead) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = a * 16 + 4 * r + lk;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int col = b * 16 + lr;
          la[a][b][r] = (row < jbn && col <= row) ? W[(long long)(base + row) * M + (base + col)] : 0.0;
        }
      }
  }
  __syncthreads();
  STAMP(3);
  if (ahead) {
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
      double fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = -Ls[0][kk * 4 + lk][i * 16 + lr];
        fb[i] = Ls[1][kk * 4 + lk][i * 16 + lr];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) la[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], la[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = a * 16 + 4 * r + lk;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int col = b * 16 + lr;
          D[row][col] = (row < jbn && col <= row) ? la[a][b][r] : 0.0;
        }
      }
    // (one wave: its LDS accesses are processed in order, no barrier between the stores above and the loads below)
    const int bad = factor_diag_wave(D, Dinv, Lv, jbn, lane);
    if (bad) {
      if (lane == 0) info[q] = base + bad;
    } else {
#pragma unroll
      for (int k = 0; k < NB / 2; ++k) {
        const int r = lane & (NB - 1), c = 2 * k + (lane >> 5);
        if (r < jbn && c <= r) Lo[(long long)(base + r) * M + (base + c)] = D[r][c];
      }
      if (lane < NB && base + NB < M) Lo[dinv_slot(base, lane, M)] = Dinv[lane];   // (only a block with rows below it needs them)
    }
#ifdef POTRF_STAMPS
    if (stamps && blockIdx.y == 0 && t == 256) stamps[6] = clock64();
#endif
    return;
  }
  if (idle) return;
#pragma unroll
  for (int kk = 0; kk < NB / 4; ++kk) {
    double fa[NS], fb[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      fa[i] = -Ls[0][kk * 4 + lk][wm * WT + i * 16 + lr];  // W - L21_i L21_j^T
      fb[i] = Ls[1][kk * 4 + lk][wn * WT + i * 16 + lr];
    }
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
      for (int b = 0; b < NS; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
  }
  STAMP(4);
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = base + ti * TS + wm * WT + a * 16 + 4 * r + lk;
      // (the next diagonal block, rows base .. base + 31 of tile (0, 0), is NOT written back: the look-ahead block of this
      // launch reads its un-updated values from W, and nothing reads it from W again -- the next launch takes it from Lo)
      if (row >= M || row < base + NB) continue;
      double* wrow = W + (long long)row * M;
#pragma unroll
      for (int b = 0; b < NS; ++b) {
        const int col = base + tj * TS + wn * WT + b * 16 + lr;
        if (col <= row) wrow[col] = acc[a][b][r];
      }
    }
  STAMP(5);
}

// A <- lower triangle of the out-of-place factor, zeros above the diagonal.
__global__ void potrf_finalize_kernel(double* __restrict__ A, int M, const double* __restrict__ Lo) {
  const int q = blockIdx.z;
  const long long o = ((long long)q * M + blockIdx.y) * M;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= M) return;
  A[o + c] = (c <= r) ? Lo[o + c] : 0.0;
}

// ---------------------------------------------------------------------------------------------- trtri
// Inverse of each TB x TB lower-triangular diagonal block (TB = 64: one column per lane of a wave): lane c solves L x = e_c by
// forward substitution with its column in REGISTERS (fully unrolled: static register indices) and the rows of L as uniform
// LDS broadcasts; the sum of row r runs over k = 0 .. r-1 for every lane (x[k] == 0 for k < c contributes exact zeros), so
// all lanes execute one instruction stream:  x_r = -(sum_{k<r} L_rk x_k) * (1 / L_rr).  A 64 x 64 base block replaces the
// 32 x 32 blocks plus the first level of GEMM merges (two launches of the latency-bound chain).
constexpr int TB = 64;
__global__ __launch_bounds__(256) void trtri_diag_kernel(const double* __restrict__ Lall, double* __restrict__ Xall, int M) {
  __shared__ __attribute__((aligned(16))) double D[TB][TB + 2];
  __shared__ double Rinv[TB];
  const int q = blockIdx.y, j = blockIdx.x * TB, jb = min(TB, M - j), t = threadIdx.x;
  const double* L = Lall + (long long)q * M * M;
  double* Xo = Xall + (long long)q * M * M;
  for (int e = t; e < TB * TB; e += blockDim.x) {
    const int r = e / TB, c = e % TB;
    D[r][c] = (r < jb && c <= r) ? L[(long long)(j + r) * M + (j + c)] : 0.0;
  }
  __syncthreads();
  if (t >= TB) return;              // (four waves fetch the block, one solves)
  const int c = t;
  Rinv[t] = 1.0 / D[t][t];          // all reciprocal pivots at once, one per lane (rows beyond the block: 1/0, never used)
  double x[TB];
  // (one wave: its LDS accesses are processed in order, no barrier needed before the broadcast reads of Rinv below)
#pragma unroll
  for (int r = 0; r < TB; ++r) {
    // four interleaved partial sums: the recurrence is latency-bound (one wave, dependent FMAs), not throughput-bound
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) {
      if ((k & 3) == 0) s0 = fma(D[r][k], x[k], s0);
      else if ((k & 3) == 1) s1 = fma(D[r][k], x[k], s1);
      else if ((k & 3) == 2) s2 = fma(D[r][k], x[k], s2);
      else s3 = fma(D[r][k], x[k], s3);
    }
    const double num = (r == c) ? 1.0 : -((s0 + s1) + (s2 + s3));
    x[r] = (r >= c) ? num * Rinv[r] : 0.0;
  }
  if (c < jb) {
#pragma unroll
    for (int r = 0; r < TB; ++r)
      if (r < jb) Xo[(long long)(j + r) * M + (j + c)] = x[r];
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

// A (Q x M x M, in place) <- its lower Cholesky factor; `scr` is a Q x M x M scratch (the out-of-place factor).
// panel_begin / panel_end (in panels of NB columns; -1 = to the end) let a caller enqueue the chain in two parts with other
// launches in between (the host needs ~10 us per launch, the device ~25 us per panel: see engine.hip u_algebra).
void launch_potrf_batched(double* A, int Q, int M, int* d_info, double* scr, hipStream_t stream, int panel_begin, int panel_end) {
  const int npanels = (M + NB - 1) / NB;
  if (panel_end < 0 || panel_end > npanels) panel_end = npanels;
  if (panel_begin == 0) HIP_TRY(hipMemsetAsync(d_info, 0, sizeof(int) * Q, stream));
  for (int pnl = panel_begin; pnl < panel_end; ++pnl) {
    const int j = pnl * NB;
    const int rem = M - j - std::min(NB, M - j);
    static const int forced = [] {   // HMOGP_POTRF_TILE=64|128: force one tile size (A/B runs)
      const char* e = getenv("HMOGP_POTRF_TILE");
      return e ? atoi(e) : 0;
    }();
    // 64 x 64 tiles while they give at most two blocks per CU (block 0's critical path 19 -> 10.4 us, phase stamps in
    // profiles/r04_potrf_phases.txt; 0.608 -> 0.546 ms at M = 1024, Q = 3); with more blocks than that the redundant diagonal-
    // block loads and panel solves of every block cost more than the shorter tile update gains (M = 2048: 1.53 -> 1.79 ms).
    const int T64 = (rem + 63) / 64;
    const int ts = forced == 64 || forced == 128 ? forced : ((long long)T64 * (T64 + 1) / 2 * Q <= 512 ? 64 : 128);
    const int T = (rem + ts - 1) / ts;
    const dim3 grid(std::max(1, T * (T + 1) / 2) + (rem > 0 ? 1 : 0), Q);
    if (ts == 128)
      hipLaunchKernelGGL(potrf_step_kernel<128>, grid, dim3(320), 0, stream, A, scr, M, j, d_info, pnl == npanels / 2 ? g_potrf_stamps : nullptr, pnl > 0 ? 1 : 0);
    else
      hipLaunchKernelGGL(potrf_step_kernel<64>, grid, dim3(320), 0, stream, A, scr, M, j, d_info, pnl == npanels / 2 ? g_potrf_stamps : nullptr, pnl > 0 ? 1 : 0);
  }
  if (panel_end == npanels) hipLaunchKernelGGL(potrf_finalize_kernel, dim3((M + 255) / 256, M, Q), dim3(256), 0, stream, A, M, scr);
}

// Linv = L^-1.  `tmp` is a Q x M x M scratch.
void launch_trtri_batched(const double* L, double* Linv, double* tmp, int Q, int M, hipStream_t stream, bool linv_is_zero) {
  const long long MM = (long long)M * M;
  if (!linv_is_zero) HIP_TRY(hipMemsetAsync(Linv, 0, sizeof(double) * MM * Q, stream));
  hipLaunchKernelGGL(trtri_diag_kernel, dim3((M + TB - 1) / TB, Q), dim3(256), 0, stream, L, Linv, M);
  // merge [[X11, 0], [X21, X22]] with X21 = -X22 * L21 * X11, block size s doubling
  for (int s = TB; s < M; s *= 2) {
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
    g.b_tri = +1;  // X11 is lower triangular
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
    h.a_tri = +1;  // X22 is lower triangular
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
  g.a_tri = -1, g.b_tri = +1;  // op(A) = Linv^T upper, op(B) = Linv lower: k >= max(i0, j0)
  g.nbatch = Q;
  g.sA = g.sB = g.sC = (long long)M * M;
  launch_gemm_f64(g, stream);
}
