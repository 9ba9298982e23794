// trsm_panel.hip -- [r6] the blocked triangular solves of the strict q(f) mode (the reference's dpotrs, svmogp_inf.py:214) as ONE
// kernel per 128-column block: the long-K update GEMM of the block and the substitution inside it, the block's 128 x 128 tile
// never leaving the chip in between.
//
//   forward  (DIR 0)   X Luu^T = V :   X[:, J] = (V[:, J] - X[:, <J] Luu[J, <J]^T) Luu[J, J]^-T      J = 0 .. M/128 - 1
//   backward (DIR 1)   A Luu   = X :   A[:, J] = (X[:, J] - A[:, >J] Luu[>J, J])   Luu[J, J]^-1      J = M/128 - 1 .. 0
//
// Round 5 ran, per 128-column block, one update GEMM (rowpass_gemm_kernel<1>, C -= A B) and FOUR substitution launches
// (trsm_diag_kernel: 32-column true substitution, one row per lane, + the right-looking update of the rest of the block from
// LDS), each of which streamed its 256 x 32 right-hand sides AND the not-yet-solved columns of the block from HBM and back:
// 640 column passes per block where 256 are needed, 197 GB per strict step at the headline size against 79 GB algorithmic --
// 52 ms at 3.8 TB/s with the matrix cores 12 % busy (profiles/r05_HS_*).
//
// Here a block of 8 waves owns one 128-row x 128-column tile.  Main loop = the forward contraction's (gemm_rowpass.hip: 128 x 128 x
// 16 block tile, wave tile 64 x 32, transposed accumulators, staged-first double buffering), K = the columns already solved.
// Then, with C = V - (product) IN THE ACCUMULATORS (lane (lr, lk) register r of acc[a][b] = row wm*64 + a*16 + lr, column
// wn*32 + b*16 + 4*lk + r: a lane's four registers of a sub-tile are four ADJACENT COLUMNS of one row), 8 sub-blocks of 16 columns
// x 4 groups of 4 columns:
//   (1) the lanes that own the group (lk == g, in the two waves that own the sub-block) solve their 4 x 4 triangle by TRUE
//       substitution in their own registers, four rows per lane (6 fused multiply-adds and 4 multiplications by the pivots'
//       precomputed reciprocals per row; the factor's entries as LDS broadcasts out of the k-major image the update needs anyway),
//       and write -x into columns 4 g .. 4 g + 3 of the main loop's row-major operand image;
//   (2) after ONE barrier every wave that still holds open columns of the tile takes k4-step g of a main-loop k-step with that
//       image as its A operand (C -= x L^T on the matrix cores; sub-tiles left of the front skipped by a wave-uniform mask).
// After the last group the accumulators hold the solved tile in the forward kernel's store layout: 16-byte row-coalesced stores.
// (A first version parked 128 x 16 sub-blocks in LDS and solved one row per lane against the 16 x 16 diagonal block: the fully
// unrolled triangle made the compiler hoist its 136 broadcast reads beside the 64 accumulators -- 167 spilled registers.)
// V is read once as the update's A operand panel (L2 / MALL-warm re-reads aside), each tile of the right-hand side is read once
// and written once: 2 x 19.7 GB per solve at the headline size.  Substitution arithmetic per row and block: 32 x (6 FMA + 4
// multiplications) instead of 4 x (496 + 32 divisions); everything else of the block's triangle rides the matrix cores.
// Numerics: still substitution against the factor itself (no block of Luu is ever inverted: an explicit inverse of a block of
// any width loses cond(block) digits); diagonal blocks are 4 wide instead of 32 and x_j = s_j * (1 / l_jj) replaces
// s_j / l_jj (one extra rounding of relative size 2^-53 on the pivot: a backward error of the same size LAPACK's own dtrsm
// kernels -- OpenBLAS stores the inverted diagonal at packing time -- commit).  tests/test_gpu_strict.py, test_gpu_ladder.py.
#include <cstdlib>

#include "common.h"
#include "rowpass.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, W = 8, NT = W * 64;
constexpr int KM_LD = 144, RM_LD = 18, TILE_DOUBLES = BK * KM_LD;   // LDS images of gemm_rowpass.hip / gemm_f64.hip
constexpr int NB = 2, WN = 32;
static_assert(128 * RM_LD <= TILE_DOUBLES, "the -x operand image fits a tile buffer");

struct Tile {
  double a[2][TILE_DOUBLES];
  double b[2][TILE_DOUBLES];
};

template <int DIR, bool STATS>
__global__ __launch_bounds__(NT, 4) void trsm_panel_kernel(TrsmPanelArgs g) {
  __shared__ __attribute__((aligned(16))) Tile lds;
  __shared__ double rds[BN];                                     // reciprocals of this block's 128 pivots
  const int batch = blockIdx.y;
  const long long n = g.n, i0 = (long long)blockIdx.x * BM;
  const int M = g.M, j0 = g.j0, ldv = g.ldv, ldl = g.ldl;
  double* __restrict__ V = g.V + (long long)batch * g.sV;
  const double* Vin = (g.Vsrc ? g.Vsrc : g.V) + (long long)batch * g.sV;   // where this block's right-hand sides are read from
  const double* __restrict__ L = g.Lsym + (long long)batch * g.sL;
  const double* __restrict__ rd = g.rdiag + (long long)batch * g.sR;
  const int kbeg = DIR == 0 ? 0 : j0 + BN, kend = DIR == 0 ? j0 : M;      // the columns this block's update contracts over

  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int wm = w >> 2, wn = w & 3;
  const int blr = 4 * (lr & 3) + (lr >> 2);     // transposed accumulators: see gemm_rowpass.hip (SWAP)
  if (t < BN) rds[t] = rd[j0 + t];              // (visible behind the first barrier below)

  f64x4 acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 5, bc = (t & 31) * 2;
  const long long arow = (i0 + ar < n) ? i0 + ar : n - 1;
  const double* pa = V + arow * ldv + ak;                       // (the solved columns: written by EARLIER launches of this solve)
  const double* pbcol = L + j0 + bc;                            // row k of the mirrored factor: pbcol + (k + bk) * ldl
  double ra[4], rb[4];
  auto loadA = [&](int k0) {
    const f64x2 x0 = *reinterpret_cast<const f64x2*>(pa + k0), x1 = *reinterpret_cast<const f64x2*>(pa + k0 + 2);
    ra[0] = x0.x, ra[1] = x0.y, ra[2] = x1.x, ra[3] = x1.y;
  };
  auto loadB = [&](int k0) {
    const double* q = pbcol + (long long)(k0 + bk) * ldl;
    const f64x2 y0 = *reinterpret_cast<const f64x2*>(q), y1 = *reinterpret_cast<const f64x2*>(q + 64);
    rb[0] = y0.x, rb[1] = y0.y, rb[2] = y1.x, rb[3] = y1.y;
  };
  auto stageA = [&](int buf) {
    double* sa = &lds.a[buf][ar * RM_LD + ak];
    *reinterpret_cast<f64x2*>(sa) = f64x2{ra[0], ra[1]};
    *reinterpret_cast<f64x2*>(sa + 2) = f64x2{ra[2], ra[3]};
  };
  auto stageB = [&](int buf) {
    double* sb = &lds.b[buf][bk * KM_LD + bc];
    *reinterpret_cast<f64x2*>(sb) = f64x2{rb[0], rb[1]};
    *reinterpret_cast<f64x2*>(sb + 64) = f64x2{rb[2], rb[3]};
  };
  double fa[2][4], fb[2][NB];
  auto frag = [&](int bufa, int bufb, int kk, int s) {
    const double* fpa = &lds.a[bufa][(wm * 64 + lr) * RM_LD + kk * 4 + lk];
    const double* fpb = &lds.b[bufb][(kk * 4 + lk) * KM_LD + wn * WN + blr];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[s][i] = fpa[i * 16 * RM_LD];
#pragma unroll
    for (int i = 0; i < NB; ++i) fb[s][i] = fpb[i * 16];
  };
  auto mm8 = [&](int s, unsigned mask) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!((mask >> b) & 1u)) continue;                     // wave-uniform (scalar) guard
        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[s][b], fa[s][a], acc[a][b], 0, 0, 0);
      }
  };

  // ---- update: acc = V[:, kbeg:kend] op(B)   (the forward contraction's main loop, gemm_rowpass.hip) ---------------------------
  int cur = 0;
  if (kbeg < kend) {
    loadA(kbeg), loadB(kbeg);
    stageA(0), stageB(0);
    if (kbeg + BK < kend) loadA(kbeg + BK), loadB(kbeg + BK);
  }
  __syncthreads();
  if (kbeg < kend) frag(0, 0, 0, 0);
#define HM_SB __builtin_amdgcn_sched_barrier(0)
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    frag(cur, cur, 1, 1);
    HM_SB;
    mm8(0, 3u);
    HM_SB;
    if (more) stageA(cur ^ 1), stageB(cur ^ 1);
    if (k0 + 2 * BK < kend) loadA(k0 + 2 * BK), loadB(k0 + 2 * BK);
    HM_SB;
    frag(cur, cur, 2, 0);
    HM_SB;
    mm8(1, 3u);
    HM_SB;
    frag(cur, cur, 3, 1);
    HM_SB;
    mm8(0, 3u);
    HM_SB;
    __syncthreads();
    if (more) frag(cur ^ 1, cur ^ 1, 0, 0);
    HM_SB;
    mm8(1, 3u);
    HM_SB;
    cur ^= 1;
  }
#undef HM_SB

  // ---- C = right-hand side - update, in the accumulators: lane (lr, lk) register r of acc[a][b] = row wm*64 + a*16 + lr,
  // ---- column wn*32 + b*16 + 4*lk + r of the tile ------------------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long long row = i0 + wm * 64 + a * 16 + lr;
    const double* srow = Vin + ((row < n) ? row : n - 1) * ldv + j0 + wn * WN + 4 * lk;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f64x2 c01 = *reinterpret_cast<const f64x2*>(srow + b * 16), c23 = *reinterpret_cast<const f64x2*>(srow + b * 16 + 2);
      acc[a][b][0] = c01.x - acc[a][b][0], acc[a][b][1] = c01.y - acc[a][b][1];
      acc[a][b][2] = c23.x - acc[a][b][2], acc[a][b][3] = c23.y - acc[a][b][3];
    }
  }

  // ---- the tile's own 128 x 128 triangle: 8 sub-blocks of 16 columns x 4 groups of 4 columns ------------------------------------
  // A lane's four accumulator registers of a sub-tile ARE four adjacent columns of one row: the group the lanes lk == g own.  Those
  // lanes solve their 4 x 4 triangle in their own registers (no parking, no transposition), write -x into columns 4 g .. 4 g + 3 of
  // the row-major operand image, and after ONE barrier every wave with open columns takes k4-step g of a main-loop k-step with
  // that image (the factor's rows 16 s .. 16 s + 15 of the block sit in the k-major image, staged one sub-block ahead).  The MFMA
  // also lands on the solved columns of the owners' sub-tile (the factor's image is not triangular): those lanes take their x
  // back from the image afterwards.
  const int s_first = DIR == 0 ? 0 : 7, sdir = DIR == 0 ? 1 : -1;
  loadB(j0 + 16 * s_first);
  stageB(0);
  int sb = 0;
  __syncthreads();
  for (int it = 0; it < 8; ++it) {
    const int s = s_first + sdir * it;
    const bool own = wn == (s >> 1);
    double* xi = &lds.a[sb][0];                                  // -x of this sub-block, [128][RM_LD]
    const double* Lb = &lds.b[sb][16 * s];                       // Lb[k * KM_LD + i] = Lsym[c0 + k][c0 + i],  c0 = j0 + 16 s
    if (it + 1 < 8) loadB(j0 + 16 * (s + sdir));                 // the next sub-block's rows of the factor: in flight meanwhile
    for (int gi = 0; gi < 4; ++gi) {
      const int gq = DIR == 0 ? gi : 3 - gi;
      if (own && lk == gq) {
        // TRUE substitution on the 4 x 4 diagonal block (pivots through their reciprocals), four rows per lane
        double l[4][4], rv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          rv[k] = rds[16 * s + 4 * gq + k];
#pragma unroll
          for (int i = 0; i < 4; ++i) l[k][i] = Lb[(4 * gq + k) * KM_LD + 4 * gq + i];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b != (s & 1)) continue;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            double x[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (DIR == 0) {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                x[k] *= rv[k];
#pragma unroll
                for (int i = k + 1; i < 4; ++i) x[i] = fma(-x[k], l[k][i], x[i]);
              }
            } else {
#pragma unroll
              for (int k = 3; k >= 0; --k) {
                x[k] *= rv[k];
#pragma unroll
                for (int i = 0; i < k; ++i) x[i] = fma(-x[k], l[k][i], x[i]);
              }
            }
            acc[a][b][0] = x[0], acc[a][b][1] = x[1], acc[a][b][2] = x[2], acc[a][b][3] = x[3];
            double* xp = &xi[(wm * 64 + a * 16 + lr) * RM_LD + 4 * gq];
            *reinterpret_cast<f64x2*>(xp) = f64x2{-x[0], -x[1]};
            *reinterpret_cast<f64x2*>(xp + 2) = f64x2{-x[2], -x[3]};
          }
        }
      }
      __syncthreads();
      // C -= x L^T for the columns still open: k4-step gq of one main-loop k-step
      unsigned mask = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int cb = wn * NB + b;
        if (DIR == 0 ? (cb > s || (cb == s && gq < 3)) : (cb < s || (cb == s && gq > 0))) mask |= 1u << b;
      }
      if (mask) {
        frag(sb, sb, gq, 0);
        mm8(0, mask);
      }
      if (own && (DIR == 0 ? (lk <= gq && gq < 3) : (lk >= gq && gq > 0))) {   // solved columns of the owners' sub-tile: x back
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b != (s & 1)) continue;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const double* xp = &xi[(wm * 64 + a * 16 + lr) * RM_LD + 4 * lk];
            const f64x2 x01 = *reinterpret_cast<const f64x2*>(xp), x23 = *reinterpret_cast<const f64x2*>(xp + 2);
            acc[a][b][0] = -x01.x, acc[a][b][1] = -x01.y, acc[a][b][2] = -x23.x, acc[a][b][3] = -x23.y;
          }
        }
      }
    }
    if (it + 1 < 8) stageB(sb ^ 1);
    sb ^= 1;
    __syncthreads();       // the next sub-block's factor rows are visible; its image buffer (read two sub-blocks ago) is free
  }

  // ---- the solved tile leaves as 16-byte stores (the forward kernel's layout) ------------------------------------------------
  if (STATS) {
    // [r6] row statistics of the solve-based q(f) (svmogp_inf.py:216-218) while the solved tile is in registers:
    //   sp += (tile) . v[columns]        v = st_vec (Luu^-1 m_q for the one-solve form: m_fd = A m = X (Luu^-1 m))
    //   sk += rowsum(tile .* K)          K = st_K, or the tile ITSELF when st_K is null (rowsum(X .* X) = rowsum(A .* K^))
    // over this block's 128 columns; partials per wave column.  The launches of a solve run one after the other on the stream, so
    // block J ADDS its partial to what the earlier blocks left there (one lane per (wave column, row) and launch: no race, fixed
    // order); trsm_stats_combine_kernel sums the four wave columns.  Replaces phase 0 of strict_rowstats_kernel (a pass over A
    // and K^: 39 GB per step at the headline size).
    const double* __restrict__ Kh = g.st_K ? g.st_K + (long long)batch * g.sV : nullptr;
    const double* __restrict__ vec = g.st_vec + (long long)batch * g.st_vecB;
    double* part = g.st_part + (long long)batch * g.st_sPart;    // [2][4][ld]
    const bool first = DIR == 0 ? j0 == 0 : j0 + BN >= M;         // the first block of this direction
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long long row = i0 + wm * 64 + a * 16 + lr;
      const double* krow = Kh ? Kh + ((row < n) ? row : n - 1) * ldv + j0 + wn * WN + 4 * lk : nullptr;
      double sp = 0.0, sk = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        double kv[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
        if (Kh) {
          const f64x2 k01 = *reinterpret_cast<const f64x2*>(krow + b * 16), k23 = *reinterpret_cast<const f64x2*>(krow + b * 16 + 2);
          kv[0] = k01.x, kv[1] = k01.y, kv[2] = k23.x, kv[3] = k23.y;
        }
        // (the lane's 4 entries of `vec` are re-read per slice: cache hits, and 16 registers less than holding all 8 across the loop)
        const double* vp = vec + (long long)(j0 + wn * WN + b * 16 + 4 * lk) * g.st_vecS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sp = fma(acc[a][b][r], vp[(long long)r * g.st_vecS], sp);
          sk = fma(acc[a][b][r], kv[r], sk);
        }
      }
      sp += __shfl_xor(sp, 16, 64), sk += __shfl_xor(sk, 16, 64);
      sp += __shfl_xor(sp, 32, 64), sk += __shfl_xor(sk, 32, 64);
      if (lk == 0 && row < n) {
        double* o = part + (long long)wn * g.st_ld + row;
        const long long ss = 4 * g.st_ld;
        o[0] = first ? sp : o[0] + sp;
        o[ss] = first ? sk : o[ss] + sk;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long long row = i0 + wm * 64 + a * 16 + lr;
    if (row >= n) continue;
    double* crow = V + row * ldv + j0 + wn * WN + 4 * lk;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      *reinterpret_cast<f64x2*>(crow + b * 16) = f64x2{acc[a][b][0], acc[a][b][1]};
      *reinterpret_cast<f64x2*>(crow + b * 16 + 2) = f64x2{acc[a][b][2], acc[a][b][3]};
    }
  }
}

// rdiag[q][j] = 1 / L[q][j][j]
__global__ __launch_bounds__(256) void rdiag_kernel(const double* __restrict__ L, long long sL, int M, double* __restrict__ out,
                                                    long long sR) {
  const int j = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
  if (j < M) out[q * sR + j] = 1.0 / L[q * sL + (long long)j * M + j];
}

// p[q][n] = sum over the 4 wave-column partials of slot 0,  c[q][n] = t2[q][n] - (sum of slot 1)      (fixed order)
__global__ __launch_bounds__(256) void trsm_stats_combine_kernel(const double* __restrict__ part, long long sPart, long long ld,
                                                                 int nparts, long long n, const double* __restrict__ t2,
                                                                 double* __restrict__ p, double* __restrict__ c, long long ldn) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int q = blockIdx.y;
  if (i >= n) return;
  const double* a = part + q * sPart + i;
  double sp = 0.0, sk = 0.0;
  for (int k = 0; k < nparts; ++k) sp += a[k * ld], sk += a[(nparts + k) * ld];
  p[q * ldn + i] = sp;
  c[q * ldn + i] = t2[q * ldn + i] - sk;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool trsm_panel_eligible(const TrsmPanelArgs& g) {
  static const bool enabled = [] {   // HMOGP_TRSM_PANEL=0: the round-5 path (update GEMM + four substitution launches per block; A/B runs)
    const char* e = getenv("HMOGP_TRSM_PANEL");
    return !(e && e[0] == '0');
  }();
  return enabled && g.n >= 1 && g.M >= BN && (g.M % BN) == 0 && (g.ldv & 1) == 0 && (g.ldl & 1) == 0 && (g.sV & 1) == 0 &&
         (g.sL & 1) == 0 && al16(g.V) && al16(g.Lsym) && (!g.Vsrc || al16(g.Vsrc)) && g.rdiag != nullptr;
}

void launch_trsm_panel(int dir, const TrsmPanelArgs& g, hipStream_t s) {
  if (g.n <= 0) return;
  dim3 grid((unsigned)((g.n + BM - 1) / BM), g.Q);
  const bool stats = g.st_part != nullptr;
  if (stats && (!g.st_vec || (g.st_K && !al16(g.st_K))))
    throw HipError{hipErrorInvalidValue, "trsm_panel: statistics need st_vec (and a 16-byte aligned st_K)", __FILE__, __LINE__};
  if (dir == 0 && stats) hipLaunchKernelGGL((trsm_panel_kernel<0, true>), grid, dim3(NT), 0, s, g);
  else if (dir == 0) hipLaunchKernelGGL((trsm_panel_kernel<0, false>), grid, dim3(NT), 0, s, g);
  else if (stats) hipLaunchKernelGGL((trsm_panel_kernel<1, true>), grid, dim3(NT), 0, s, g);
  else hipLaunchKernelGGL((trsm_panel_kernel<1, false>), grid, dim3(NT), 0, s, g);
}

void launch_rdiag(const double* L, long long sL, int M, int Q, double* out, long long sR, hipStream_t s) {
  hipLaunchKernelGGL(rdiag_kernel, dim3((M + 255) / 256, Q), dim3(256), 0, s, L, sL, M, out, sR);
}

void launch_trsm_stats_combine(const double* part, long long sPart, long long ld, int nparts, long long n, int Q, const double* t2,
                               double* p, double* c, long long ldn, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(trsm_stats_combine_kernel, dim3((unsigned)((n + 255) / 256), Q), dim3(256), 0, s, part, sPart, ld, nparts, n, t2, p,
                     c, ldn);
}
