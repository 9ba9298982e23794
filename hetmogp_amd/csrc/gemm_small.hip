// gemm_small.hip -- FP64-MFMA GEMM with 64 x 64 block tiles for the replicated M x M algebra of the svmogp_inf path
// (gfx950): C[b] = alpha * op(A[b]) op(B[b]), row-major C, every operand layout of gemm_f64.hip.
//
// Why a second tile size: the ~25 products of one evaluation that are M x M x M with M = 512..2048 (K_uu^-1 S, K_uu^-1 S
// K_uu^-1, H K_uu^-1, K_uu^-1 H K_uu^-1, G S K_uu^-1, dL/dS L, S = L L^T, the upper levels of the triangular inverse) give
// the 128 x 128 kernel only 64 x Q = 192 blocks at M = 1024, Q = 3: one block on three quarters of the CUs, i.e. ONE wave
// per SIMD -- and one wave per SIMD reaches half the FP64-MFMA rate (tools/probes/probe_coissue.hip).  They run at
// 0.35-0.43 MFMA utilisation, ~155 us each, in a dependent chain that is the Amdahl term of a multi-GPU step.  With
// 64 x 64 tiles the same product is 768 blocks of 4 waves (wave tile 32 x 32 = 2 x 2 MFMA tiles): three blocks on every
// CU, three waves per SIMD.  Operand traffic per flop doubles, but these operands are 8 MB matrices that live in L2 / MALL.
//
// LDS images (one 64 x 16 operand tile = 1024 doubles): k-major [k][80], row-major [row][18] -- 80*8 B = 160 dwords == 32
// (mod 64 banks) puts the two k-rows of each 32-lane half of a ds_read_b64 on disjoint banks, as 144 does for 128 columns.
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, NTH = 256;
constexpr int KM_LD = 80, RM_LD = 18, TILE_D = TK * KM_LD;   // 1280 doubles >= 64 * 18 = 1152
static_assert(TM * RM_LD <= TILE_D, "both LDS images fit one buffer");

struct Tile {
  double a[2][TILE_D];
  double b[2][TILE_D];
};

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NTH, 3) void gemm_small_kernel(GemmArgs g, int tiles_n, int a_vec, int b_vec) {
  __shared__ __attribute__((aligned(16))) Tile lds;
  int ti, tj;
  if (g.lower_only) {   // lower tiles only (tile_col <= tile_row), enumerated row by row: v = ti (ti + 1) / 2 + tj
    const int v = blockIdx.x;
    ti = (int)((sqrt(8.0 * (double)v + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= v) ++ti;
    while (ti * (ti + 1) / 2 > v) --ti;
    tj = v - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x / tiles_n, tj = blockIdx.x - ti * tiles_n;
  }
  const int i0 = ti * TM, j0 = tj * TN, batch = blockIdx.z;
  const long long ob = blockIdx.y;
  const double* __restrict__ A = g.A + (long long)batch * g.sA + ob * g.oA;
  const double* __restrict__ B = g.B + (long long)batch * g.sB + ob * g.oB;
  double* __restrict__ C = g.C + (long long)batch * g.sC + ob * g.oC;
  // triangular operands (GemmArgs::a_tri / b_tri): trim the k-loop to the products that can be non-zero
  int wlo = 0, whi = g.K;
  if (g.a_tri > 0) whi = min(whi, i0 + TM);
  if (g.b_tri < 0) whi = min(whi, j0 + TN);
  if (g.a_tri < 0) wlo = max(wlo, i0);
  if (g.b_tri > 0) wlo = max(wlo, j0);
  wlo = min(wlo & ~(TK - 1), whi);
  whi = min(g.K, (whi + TK - 1) & ~(TK - 1));

  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  // operand streams: 4 doubles per thread, operand and k-step
  //   k-major  [k][col]: thread -> k row t/16, columns (t%16)*4 .. +3        (512 contiguous bytes per k row)
  //   row-major [row][k]: thread -> row t/4, k's (t%4)*4 .. +3
  const int kr = t >> 4, kc = (t & 15) * 4, rr = t >> 2, rk = (t & 3) * 4;
  const double* pa = A_KMAJOR ? A + (long long)kr * g.lda + i0 + kc : A + (long long)(i0 + rr) * g.lda + rk;
  const double* pb = B_KMAJOR ? B + (long long)kr * g.ldb + j0 + kc : B + (long long)(j0 + rr) * g.ldb + rk;
  double ra[4], rb[4];
  auto load = [&](int k0) {
    const double* qa = A_KMAJOR ? pa + (long long)k0 * g.lda : pa + k0;
    const double* qb = B_KMAJOR ? pb + (long long)k0 * g.ldb : pb + k0;
    if (a_vec) {   // (uniform) 16-byte aligned operand: two dwordx4 loads; else four dwordx2 (the statistic bundle's H_q)
      const f64x2 x0 = *reinterpret_cast<const f64x2*>(qa), x1 = *reinterpret_cast<const f64x2*>(qa + 2);
      ra[0] = x0.x, ra[1] = x0.y, ra[2] = x1.x, ra[3] = x1.y;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = qa[i];
    }
    if (b_vec) {
      const f64x2 y0 = *reinterpret_cast<const f64x2*>(qb), y1 = *reinterpret_cast<const f64x2*>(qb + 2);
      rb[0] = y0.x, rb[1] = y0.y, rb[2] = y1.x, rb[3] = y1.y;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = qb[i];
    }
  };
  auto stage = [&](int buf) {
    double* sa = A_KMAJOR ? &lds.a[buf][kr * KM_LD + kc] : &lds.a[buf][rr * RM_LD + rk];
    double* sb = B_KMAJOR ? &lds.b[buf][kr * KM_LD + kc] : &lds.b[buf][rr * RM_LD + rk];
    *reinterpret_cast<f64x2*>(sa) = f64x2{ra[0], ra[1]};
    *reinterpret_cast<f64x2*>(sa + 2) = f64x2{ra[2], ra[3]};
    *reinterpret_cast<f64x2*>(sb) = f64x2{rb[0], rb[1]};
    *reinterpret_cast<f64x2*>(sb + 2) = f64x2{rb[2], rb[3]};
  };
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  // stage-first software pipeline (see gemm_rowpass.hip): step k + 1 goes from registers to the other buffer before the MFMA
  // block of step k, the registers are refilled with step k + 2 at once
  int cur = 0;
  if (wlo < whi) {
    load(wlo);
    stage(0);
    if (wlo + TK < whi) load(wlo + TK);
  }
  __syncthreads();
  for (int k0 = wlo; k0 < whi; k0 += TK) {
    if (k0 + TK < whi) stage(cur ^ 1);
    if (k0 + 2 * TK < whi) load(k0 + 2 * TK);
#pragma unroll
    for (int kk = 0; kk < TK / 4; ++kk) {
      const double* fpa = A_KMAJOR ? &lds.a[cur][(kk * 4 + lk) * KM_LD + wm * 32 + lr] : &lds.a[cur][(wm * 32 + lr) * RM_LD + kk * 4 + lk];
      const double* fpb = B_KMAJOR ? &lds.b[cur][(kk * 4 + lk) * KM_LD + wn * 32 + lr] : &lds.b[cur][(wn * 32 + lr) * RM_LD + kk * 4 + lk];
      double fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = fpa[i * 16 * (A_KMAJOR ? 1 : RM_LD)];
        fb[i] = fpb[i * 16 * (B_KMAJOR ? 1 : RM_LD)];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }
  // D fragment of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const double alpha = g.alpha;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 32 + a * 16 + 4 * r + lk) * g.ldc + j0 + wn * 32 + lr;
#pragma unroll
      for (int b = 0; b < 2; ++b) crow[b * 16] = alpha * acc[a][b][r];
    }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// Full 64 x 64 tiles, a k-extent that is a multiple of 16, no accumulation into C, no K split, no ragged last batch, no
// row-pass extras.  Operands that are not 16-byte aligned (odd strides / offsets) are read with 8-byte loads.
// lower_only (square products): only the tiles on or below the diagonal are launched -- 136 of 256 at M = 1024.
bool gemm_small_eligible(const GemmArgs& g) {
  if (g.lower_only && g.M != g.N) return false;
  if (g.role != 0 || g.beta != 0.0 || g.ksplit != 1 || g.kscale || g.win || g.fs_part) return false;
  if ((g.M_last && g.M_last != g.M) || (g.N_last && g.N_last != g.N) || (g.K_last && g.K_last != g.K)) return false;
  if ((g.M % TM) || (g.N % TN) || (g.K % TK)) return false;
  return true;
}

void launch_gemm_small(const GemmArgs& g, hipStream_t stream) {
  const int tiles_m = g.M / TM, tiles_n = g.N / TN;
  dim3 grid(g.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n, g.nouter, g.nbatch);
  const int a_vec = !((g.lda & 1) || (g.sA & 1) || (g.oA & 1) || !al16(g.A));
  const int b_vec = !((g.ldb & 1) || (g.sB & 1) || (g.oB & 1) || !al16(g.B));
  if (g.a_kmajor && g.b_kmajor)
    hipLaunchKernelGGL((gemm_small_kernel<true, true>), grid, dim3(NTH), 0, stream, g, tiles_n, a_vec, b_vec);
  else if (g.a_kmajor && !g.b_kmajor)
    hipLaunchKernelGGL((gemm_small_kernel<true, false>), grid, dim3(NTH), 0, stream, g, tiles_n, a_vec, b_vec);
  else if (!g.a_kmajor && g.b_kmajor)
    hipLaunchKernelGGL((gemm_small_kernel<false, true>), grid, dim3(NTH), 0, stream, g, tiles_n, a_vec, b_vec);
  else
    hipLaunchKernelGGL((gemm_small_kernel<false, false>), grid, dim3(NTH), 0, stream, g, tiles_n, a_vec, b_vec);
}
