// gemm_rowpass.hip -- the two row-pass contractions of the svmogp_inf path as SPECIALISED FP64-MFMA kernels (gfx950):
//   forward   P~ = K^ C_q  (+ fused row statistics)      n x M x M, A row-major, B k-major        svmogp_inf.py:212-218
//   Gram      H_q += K^T diag(beta) K^ (lower tiles)     M x M x n, both operands k-major, split over n   :145-147
// Same block tile (128 x 128 x 16), LDS images, XCD-aware block order and fused epilogue as the general kernel
// (gemm_f64.hip), but
//   * EIGHT waves per block, wave tile 64 x 32 (8 accumulators, <= 128 VGPRs): two blocks per CU = FOUR waves per SIMD.
//     One wave per SIMD reaches half the FP64-MFMA rate, two reach all of it (tools/probes/probe_coissue.hip): with two
//     waves per SIMD every barrier / waitcnt stall of one idles half the pipe; four leave slack
//     (tools/probes/probe_gemm8.hip: 67.8 vs 64.0 TFLOP/s on the bare loop);
//   * a branch-free main loop: full 128-column tiles and a k-extent that is a multiple of 16 are REQUIRED of the
//     inducing dimension (gemm_rowpass_eligible; anything else takes the general kernel), ragged ROWS are handled by
//     clamping the row index once per thread (forward) or by a zero k-scale (Gram), so a k-step is: 2-4 vector loads,
//     32 MFMAs fed by 12 ds_read, 2-4 ds_write, one barrier -- no per-step bookkeeping.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, W = 8, NT = W * 64;
constexpr int KM_LD = 144, RM_LD = 18, TILE_DOUBLES = BK * KM_LD;   // LDS images: see gemm_f64.hip
constexpr int NB = 2, WN = 32, NWN = 4;                              // wave tile 64 x 32 = 4 x 2 sub-tiles; 4 wave columns

struct Tile {
  double a[2][TILE_DOUBLES];
  double b[2][TILE_DOUBLES];
};

// Diagonal tiles of the lower-only Gram: wave -> wave tile (wm, wn): (1,0) (1,1) (0,0) (1,2) (0,2) (0,3) (1,3) (0,1), i.e.
// 8, 8, 7, 7, 0, 0, 3, 3 needed sub-tiles: the two waves of every SIMD (w and w + 4) carry 8, 8, 10, 10 instead of 16.
__device__ __forceinline__ int diag_wm(int w) { return (0x4B >> w) & 1; }
__device__ __forceinline__ int diag_wn(int w) { return (0x7E84 >> (2 * w)) & 3; }

// One 128 x 128 tile of the contraction: main loop + epilogue (everything the kernel does once it knows its tile).
template <int ROLE, bool FOLD = false>
__device__ __forceinline__ void rowpass_tile(const GemmArgs& g, int tiles_n, int ti, int tj, int split, Tile& lds, double* epi_a,
                                             int* hwinfo = nullptr) {
  const int batch = blockIdx.z;
  const int M = g.M, N = g.N, K = g.K;
  const int i0 = ti * BM, j0 = tj * BN;
  if (i0 >= M || j0 >= N) return;
  int wlo = 0;
  if (ROLE == 1 && g.b_tri > 0) wlo = j0;                      // triangular fold of C: op(B)[k][j] == 0 for k < j
  const int ksteps = (K - wlo + BK - 1) / BK;
  const int per = (ksteps + g.ksplit - 1) / g.ksplit;
  const int kbeg = wlo + split * per * BK;
  const int kend = min(K, kbeg + per * BK);

  const double* __restrict__ A = g.A + (long long)batch * g.sA;
  const double* __restrict__ B = g.B + (long long)batch * g.sB;
  const double* __restrict__ S = (ROLE == 2) ? g.kscale + (long long)batch * g.sS : nullptr;
  double* __restrict__ C = g.C + (long long)batch * g.sC + (long long)split * g.sSplit;
  double* fs_part = (ROLE == 1 && g.fs_part) ? g.fs_part + (long long)batch * g.fs_sPart : nullptr;

  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  int wm = w >> 2, wn = w & 3;
  // [r5] Triangular fold: inside the DIAGONAL 128-row block of C's fold (k in [j0, j0 + 128)) the wave column wn only meets
  // non-zero entries from k = j0 + 32 wn on (op(B)[k][j] == 0 for k < j), so its MFMAs of the earlier k-steps are skipped: 8, 6,
  // 4, 2 of the block's 8 k-steps for wn = 0 .. 3 (products with exact zeros: the accumulators keep their bits).  Waves w and
  // w + 4 share a SIMD, so the upper four take the wave columns in REVERSE order: every SIMD carries 10 of 16 wave-steps.
  // (FOLD: the paired kernel only -- the plain forward instantiation keeps its code)
  if (FOLD && w >= 4) wn = 3 - wn;
  const int klive = FOLD ? j0 + 32 * wn : 0;                    // first k-step whose MFMAs this wave issues
  unsigned sub = 0xFFu;                                         // bit a*2 + b: sub-tile (a, b) of the wave tile is computed
  const bool diag = ROLE == 2 && ti == tj && g.lower_only;
  // Diagonal tiles: the four (w, w + 4) wave pairs carry 8, 8, 10, 10 of the 36 needed sub-tiles, and a pair shares a SIMD,
  // so ONE block loads the CU's four SIMDs 8 : 8 : 10 : 10.  Two blocks are resident per CU: the one in the upper wave slots
  // takes the pair types rotated by two (10 : 10 : 8 : 8 by PHYSICAL SIMD), which makes it 18 on every SIMD -- 9/16 of a
  // full tile's MFMA time instead of 10/16.  HW_REG_HW_ID (tools/probes/probe_hwid.hip, gfx950): bits 5:4 = SIMD, bits 3:0 =
  // wave slot on that SIMD (this kernel's 4 waves per SIMD sit in slots {0,1} / {2,3} by block).  Only the BALANCE depends on
  // that reading: which wave computes which sub-tile does not change any value, and if the pairs are not found on four
  // distinct SIMDs the static assignment is used.
  if (diag && hwinfo) {
    if (lane == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
      hwinfo[w] = (int)((hw >> 4) & 3u);
      if (w == 0) hwinfo[8] = (int)((hw >> 1) & 1u);                              // upper or lower pair of wave slots
    }
  }
  // The forward contraction accumulates the TRANSPOSED sub-tiles (MFMA operands swapped, B fragment columns permuted): lane
  // (lr, lk) register r of acc[a][b] holds P~[wm*64 + a*16 + lr][wn*32 + b*16 + 4*lk + r] -- a lane owns 4 adjacent columns
  // of ONE row, so the fused row statistics are in-lane sums + two shuffles and P~ leaves as 16-byte stores.
  constexpr bool SWAP = (ROLE == 1);
  const int blr = SWAP ? 4 * (lr & 3) + (lr >> 2) : lr;

  f64x4 acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  // ---- operand streams: 4 doubles per thread, operand and k-step ------------------------------------------------------
  // forward: A row-major (thread: row t/4, k's (t%4)*4..+3; rows beyond the matrix read the last row, never stored),
  //          B k-major   (thread: k row t/32, column pairs (t%32)*2 and +64)
  // Gram:    A and B k-major over the same rows (thread: k row t/32 ...); rows beyond the range get the k-scale 0
  const int ar = t >> 2, ak = (t & 3) * 4, bk = t >> 5, bc = (t & 31) * 2;
  const double* pa;
  const double* pb;
  if (ROLE == 1) {
    pa = A + (long long)min(i0 + ar, M - 1) * g.lda + ak;
    pb = B + (long long)bk * g.ldb + j0 + bc;
  } else {
    pa = A + i0 + bc;
    pb = B + j0 + bc;
  }
  double ra[4], rb[4], ks = 1.0;
  auto load = [&](int k0) {
    if (ROLE == 1) {
      const f64x2 x0 = *reinterpret_cast<const f64x2*>(pa + k0), x1 = *reinterpret_cast<const f64x2*>(pa + k0 + 2);
      ra[0] = x0.x, ra[1] = x0.y, ra[2] = x1.x, ra[3] = x1.y;
      const double* q = pb + (long long)k0 * g.ldb;
      const f64x2 y0 = *reinterpret_cast<const f64x2*>(q), y1 = *reinterpret_cast<const f64x2*>(q + 64);
      rb[0] = y0.x, rb[1] = y0.y, rb[2] = y1.x, rb[3] = y1.y;
    } else {
      const int row = k0 + bk;
      const long long off = (long long)min(row, K - 1) * g.lda;
      const f64x2 x0 = *reinterpret_cast<const f64x2*>(pa + off), x1 = *reinterpret_cast<const f64x2*>(pa + off + 64);
      ra[0] = x0.x, ra[1] = x0.y, ra[2] = x1.x, ra[3] = x1.y;
      const f64x2 y0 = *reinterpret_cast<const f64x2*>(pb + off), y1 = *reinterpret_cast<const f64x2*>(pb + off + 64);
      rb[0] = y0.x, rb[1] = y0.y, rb[2] = y1.x, rb[3] = y1.y;
      ks = (row < kend) ? S[row] : 0.0;
    }
  };
  // (the k-scale is applied at stage time, after the MFMA block: a multiply at load time makes the compiler wait for the
  // loads in front of the MFMAs and exposes the HBM latency every k-step)
  auto stage = [&](int buf) {
    if (ROLE == 1) {
      double* sa = &lds.a[buf][ar * RM_LD + ak];
      *reinterpret_cast<f64x2*>(sa) = f64x2{ra[0], ra[1]};
      *reinterpret_cast<f64x2*>(sa + 2) = f64x2{ra[2], ra[3]};
    } else {
      double* sa = &lds.a[buf][bk * KM_LD + bc];
      *reinterpret_cast<f64x2*>(sa) = f64x2{ra[0], ra[1]};
      *reinterpret_cast<f64x2*>(sa + 64) = f64x2{ra[2], ra[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] *= ks;
    }
    double* sb = &lds.b[buf][bk * KM_LD + bc];
    *reinterpret_cast<f64x2*>(sb) = f64x2{rb[0], rb[1]};
    *reinterpret_cast<f64x2*>(sb + 64) = f64x2{rb[2], rb[3]};
  };
  // MFMA operands ("fragments") of one k4-step: 4 A values + NB B values per lane, read from the LDS images into register set s
  double fa[2][4], fb[2][NB];
  auto frag = [&](int buf, int kk, int s) {
    const double* fpa = (ROLE == 1) ? &lds.a[buf][(wm * 64 + lr) * RM_LD + kk * 4 + lk] : &lds.a[buf][(kk * 4 + lk) * KM_LD + wm * 64 + lr];
    const double* fpb = &lds.b[buf][(kk * 4 + lk) * KM_LD + wn * WN + blr];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[s][i] = fpa[i * 16 * (ROLE == 1 ? RM_LD : 1)];
#pragma unroll
    for (int i = 0; i < NB; ++i) fb[s][i] = fpb[i * 16];
  };
  auto mm8 = [&](int s) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (ROLE == 2 && !((sub >> (a * NB + b)) & 1u)) continue;   // wave-uniform (scalar) guard; all set off the diagonal
        acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f64_16x16x4f64(fb[s][b], fa[s][a], acc[a][b], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s][a], fb[s][b], acc[a][b], 0, 0, 0);
      }
  };

  // Software pipeline: the operands of step k + 1 are in registers when step k starts; they are staged into the other LDS
  // buffer BEFORE the MFMA block of step k (so the ds_write latency is covered by it and the barrier follows the last MFMA
  // directly) and the registers are refilled at once with step k + 2.
  int cur = 0;
  if (kbeg < kend) {
    load(kbeg);
    stage(0);
    if (kbeg + BK < kend) load(kbeg + BK);
  }
  __syncthreads();
  if (diag) {
    int role = w;                                               // static assignment: pair type w & 3, half w >> 2
    if (hwinfo) {
      bool ok = true;
      unsigned seen = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ok = ok && hwinfo[i] == hwinfo[i + 4];
        seen |= 1u << hwinfo[i];
      }
      if (ok && seen == 0xFu) role = ((hwinfo[w] + 2 * hwinfo[8]) & 3) + (w & 4);
    }
    role = __builtin_amdgcn_readfirstlane(role);
    wm = diag_wm(role), wn = diag_wn(role);
    sub = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (wn * NB + b <= wm * 4 + a) sub |= 1u << (a * NB + b);
  }
  // [r4] Main loop with the fragments ONE k4-step AHEAD OF THE MFMAs AND CARRIED ACROSS THE BARRIER: the fragments of the next
  // tile's first k4-step are read right behind the barrier, in front of the last 8 MFMAs of the current tile, and the staging
  // of step k + 1 / the global loads of step k + 2 sit behind the first 8 MFMAs -- a wave never waits for LDS (or for the
  // ds_write -> ds_read turn-around at the top of a step) without 8 MFMAs of its own in flight.  Ablation of the bare loop
  // (tools/probes/probe_gemm_abl.hip, profiles/r04_gemm_ablation.txt): 66.3 -> 70.8 TFLOP/s; the same loop without any
  // barrier (racy) 72.5, MFMAs alone in this tile structure 73.6.  Same operations on the same values in the same order per
  // accumulator: results are bit-identical to the round-3 loop.
  // The Gram keeps the round-3 loop (fragments read inside the step): the carried fragments cost 12 registers, the Gram's 112
  // allocated registers are what lets the column statistics run BESIDE it (one 64-register guest wave per SIMD), and alone it
  // gains nothing (62.2 -> 62.3 TFLOP/s; in the step 39.4 -> 42.9 ms without the guest).
  const bool live = ROLE == 1 || sub != 0;                     // (waves of a diagonal tile without a sub-tile idle through it)
  if (ROLE == 2) {
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      if (k0 + BK < kend) stage(cur ^ 1);
      if (k0 + 2 * BK < kend) load(k0 + 2 * BK);
      if (live) {
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
          frag(cur, kk, 0);
          mm8(0);
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  if (ROLE == 1 && kbeg < kend && (!FOLD || kbeg >= klive)) frag(0, 0, 0);
#define HM_SB __builtin_amdgcn_sched_barrier(0)
  for (int k0 = kbeg; ROLE == 1 && k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    const bool lv = !FOLD || k0 >= klive;                      // (wave-uniform; always true outside the fold's diagonal block)
    if (lv) {
      frag(cur, 1, 1);
      HM_SB;
      mm8(0);
      HM_SB;
    }
    if (more) stage(cur ^ 1);
    if (k0 + 2 * BK < kend) load(k0 + 2 * BK);
    HM_SB;
    if (lv) {
      frag(cur, 2, 0);
      HM_SB;
      mm8(1);
      HM_SB;
      frag(cur, 3, 1);
      HM_SB;
      mm8(0);
      HM_SB;
    }
    __syncthreads();
    if (more && (!FOLD || k0 + BK >= klive)) frag(cur ^ 1, 0, 0);
    if (lv) {
      HM_SB;
      mm8(1);
      HM_SB;
    }
    cur ^= 1;
  }
#undef HM_SB

  // ---- epilogue: D fragment of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg --------------------------
  if (ROLE == 1 && fs_part && g.fs_sq) {
    // [r5] strict q(f): the only statistic is rowsum(T .* T) of the product itself -- in-lane squares of the lane's 8 values per
    // slice + the two shuffles over the four lanes of a row; same partial layout as `c` below (slot 1 of [stat][4 tiles_n][M])
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double sq = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = fma(acc[a][b][r], acc[a][b][r], sq);
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      const int rl = wm * 64 + a * 16 + lr;
      if (lk == 0 && i0 + rl < M) fs_part[((long long)NWN * tiles_n + NWN * tj + wn) * M + (i0 + rl)] = sq;
    }
    if (!g.store_c) return;
  } else if (ROLE == 1 && fs_part) {
    // Fused row statistics (GemmArgs::fs_*): lane (lr, lk) owns row rl = wm*64 + a*16 + lr and the column quads
    // wn*32 + b*16 + 4*lk + (0..3), b < 2, of slice a; it re-reads exactly its own 8 values of the K^ tile per slice
    // (L2/MALL-warm) with LDS-DMA loads (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16*l) into a
    // wave-private double buffer -- no VGPRs, no block barriers.  Slices 0 and 1 are issued up front, slice a + 2 as soon as
    // slice a has been consumed (into the buffer it frees): every slice but the first has two consumption periods to land
    // in.  The waits count on loads returning in order: s_waitcnt vmcnt(4) with [slice a | stores | slice a + 1] outstanding
    // can only be satisfied once slice a has landed (were one of its 4 loads outstanding, so were all 4 of slice a + 1).
    // [r5] TWO statistics (p = K^ a, c = rowsum(P~ .* K^)).  Their r2-weighted twins (only ever consumed as the scalar sl of the
    // lengthscale gradient) moved to colstats_kernel, which forms E_nm and x_n - z_m anyway: no per-element distances, no
    // `hyper` branch and no inputs staged here any more.
    constexpr int NCH = 2 * NB, WSTAGE = 2 * NCH * 128;
    double* flat = &lds.a[0][0];                  // Tile = 4 * TILE_DOUBLES contiguous doubles
    double* stg = flat + w * WSTAGE;              // [2 buffers][NCH chunks][64 lanes][2]
    static_assert(W * WSTAGE <= 4 * TILE_DOUBLES, "epilogue scratch fits the tile buffers");
    const int colq = j0 + wn * WN + 4 * lk;       // + b*16 (+ 2h)
    const double* __restrict__ Ke = g.fs_k ? g.fs_k + (long long)batch * g.sA : A;   // (strict q(f): K^ is not this product's A operand)
    auto issue = [&](int a) {
      const int grow = min(i0 + wm * 64 + a * 16 + lr, M - 1);
      const double* src = Ke + (long long)grow * g.lda + colq;
      double* dst = stg + (a & 1) * (NCH * 128);
#pragma unroll
      for (int c = 0; c < NCH; ++c)               // chunk c = 2*b + h
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (c >> 1) * 16 + (c & 1) * 2),
                                         (void __attribute__((address_space(3)))*)(dst + c * 128), 16, 0, 0);
    };
    issue(0);
    issue(1);
    if (t < 128) epi_a[t] = g.fs_a[(long long)batch * g.fs_sA + j0 + t];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a < 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // slice a has landed; slice a + 1 may still be in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int rl = wm * 64 + a * 16 + lr;
      double sp = 0.0, sc = 0.0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int cl = wn * WN + b * 16 + 4 * lk;  // this lane's 4 adjacent columns of sub-tile b (within the tile)
        const double* kp = stg + (a & 1) * (NCH * 128) + (2 * b) * 128 + 2 * lane;
        const f64x2 k01 = *reinterpret_cast<const f64x2*>(kp), k23 = *reinterpret_cast<const f64x2*>(kp + 128);
        const double kv[4] = {k01.x, k01.y, k23.x, k23.y};
        const f64x2 a01 = *reinterpret_cast<const f64x2*>(epi_a + cl), a23 = *reinterpret_cast<const f64x2*>(epi_a + cl + 2);
        const double av[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sp += kv[r] * av[r];
          sc += acc[a][b][r] * kv[r];
        }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {           // the four lanes (lk) that share this row
        sp += __shfl_xor(sp, o, 64);
        sc += __shfl_xor(sc, o, 64);
      }
      if (lk == 0 && i0 + rl < M) {                  // partial of (column tile, wave column): [stat][4 * tiles_n][M]
        double* o = fs_part + ((long long)(NWN * tj + wn)) * M + (i0 + rl);
        const long long ss = (long long)NWN * tiles_n * M;
        o[0] = sp, o[ss] = sc;
      }
      if (a < 2) {                                   // slice a is consumed: its buffer takes slice a + 2
        asm volatile("" ::: "memory");
        issue(a + 2);
      }
    }
    if (!g.store_c) return;
  }
  if (SWAP) {  // lane (lr, lk): row wm*64 + a*16 + lr, columns wn*32 + b*16 + 4*lk + (0..3)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = i0 + wm * 64 + a * 16 + lr;
      if (row >= M) continue;
      double* crow = C + (long long)row * g.ldc + j0 + wn * WN + 4 * lk;
      if (g.c_sub) {   // [r5] C -= op(A) op(B): the 128-column updates of the strict mode's blocked triangular solves
        // (c_src: the columns' FIRST touch reads them from the matrix the solve started from -- no copy of it beforehand)
        const double* srow = g.c_src ? g.c_src + (long long)batch * g.sC + (long long)row * g.ldc + j0 + wn * WN + 4 * lk : crow;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const f64x2 c01 = *reinterpret_cast<const f64x2*>(srow + b * 16), c23 = *reinterpret_cast<const f64x2*>(srow + b * 16 + 2);
          *reinterpret_cast<f64x2*>(crow + b * 16) = f64x2{c01.x - acc[a][b][0], c01.y - acc[a][b][1]};
          *reinterpret_cast<f64x2*>(crow + b * 16 + 2) = f64x2{c23.x - acc[a][b][2], c23.y - acc[a][b][3]};
        }
        continue;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        *reinterpret_cast<f64x2*>(crow + b * 16) = f64x2{acc[a][b][0], acc[a][b][1]};
        *reinterpret_cast<f64x2*>(crow + b * 16 + 2) = f64x2{acc[a][b][2], acc[a][b][3]};
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* crow = C + (long long)(i0 + wm * 64 + a * 16 + 4 * r + lk) * g.ldc + j0 + wn * WN + lr;
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if ((sub >> (a * NB + b)) & 1u) crow[b * 16] = acc[a][b][r];   // (diagonal tiles: only what lies on or below the diagonal)
    }
}

template <int ROLE, bool PAIR>
__device__ __forceinline__ void rowpass_block(const GemmArgs& g, int tiles_n, int ntiles, Tile& lds, double* epi_a,
                                              int* hwinfo = nullptr) {

  // ---- which tile / batch / k-range (block b is observed to run on XCD b % 8: speed only) ----------------------------
  int v = blockIdx.x, split = 0;
  int ti, tj;
  if (ROLE == 2 && g.lower_only) {
    // Lower tiles of the Gram, in TWO phases of equal-duration blocks.  The tiles of one K range stream the same operand rows,
    // and they only share them through the XCD's L2 if they START together: blocks of unequal duration (a diagonal tile does
    // ~0.6 of the work) spread the start times of everything behind them and the sharing is lost (measured: 36 mixed tiles
    // per range run as slowly as 36 full ones).  So: first all strictly-lower tiles, range-major, all tiles of one K range on
    // ONE XCD (block b is observed to run on XCD b % 8: speed only); then the diagonal tiles, which share nothing (tile
    // (d, d) reads panel d only).
    const int T = tiles_n, noff = T * (T - 1) / 2, ks8 = (g.ksplit > 1) ? ((g.ksplit + 7) / 8) * 8 : 1, gx1 = noff * ks8;
    if (v < gx1) {
      int u = v;                                          // strictly-lower index: u = ti (ti - 1) / 2 + tj, tj < ti
      if (g.ksplit > 1) {
        const int xcd = v & 7, idx = v >> 3;
        split = (idx / noff) * 8 + xcd;
        u = idx % noff;
      }
      ti = (int)((sqrt(8.0 * (double)u + 1.0) + 1.0) * 0.5);
      while (ti * (ti + 1) / 2 <= u) ++ti;
      while (ti * (ti - 1) / 2 > u) --ti;
      tj = u - ti * (ti - 1) / 2;
    } else {
      const int u = v - gx1;
      split = u / T;
      ti = tj = u - split * T;
    }
    if (split >= g.ksplit) return;
  } else {
    if (g.ksplit > 1) {               // all tiles of one K range on ONE XCD: they stream the same operand rows concurrently
      const int xcd = v & 7, idx = v >> 3;
      split = (idx / ntiles) * 8 + xcd;
      v = idx % ntiles;
      if (split >= g.ksplit) return;
    } else if ((ntiles & 7) == 0) {   // a contiguous range of tiles per XCD: the column tiles of a row panel share its A panel
      const int cpx = ntiles >> 3;
      v = (v & 7) * cpx + (v >> 3);
    }
    const int tcols = PAIR ? tiles_n / 2 : tiles_n;
    ti = v / tcols;
    tj = v - ti * tcols;
  }
  rowpass_tile<ROLE, PAIR>(g, tiles_n, ti, tj, split, lds, epi_a, hwinfo);
  // Triangular fold (b_tri > 0, the E-step's / predict_f's forward): the k-loop of column tile j starts at j, so the tiles of
  // a row panel do 8, 7, ... 1 eighths of a full tile's work.  In paired mode a block takes column tiles j and
  // tiles_n - 1 - j one after the other: every block does (tiles_n + 1) / tiles_n of a full tile -- equal durations.
  if (PAIR) {
    __syncthreads();                 // the epilogue of the first tile used the tile buffers as scratch
    rowpass_tile<ROLE, PAIR>(g, tiles_n, ti, tiles_n - 1 - tj, split, lds, epi_a);
  }
}

// ROLE 1 = forward, ROLE 2 = weighted Gram (names the instantiation in profiles, like gemm_f64_kernel's ROLE)
template <int ROLE>
__global__ __launch_bounds__(NT, 4) void rowpass_gemm_kernel(GemmArgs g, int tiles_n, int ntiles) {
  __shared__ __attribute__((aligned(16))) Tile lds;
  __shared__ __attribute__((aligned(16))) double epi_a[ROLE == 1 ? 128 : 2];
  __shared__ int hwinfo[ROLE == 2 ? 16 : 1];
  rowpass_block<ROLE, false>(g, tiles_n, ntiles, lds, epi_a, (ROLE == 2 && g.diag_balance) ? hwinfo : nullptr);
}
// the forward contraction against the triangular fold of C, two column tiles per block (see rowpass_block)
__global__ __launch_bounds__(NT, 4) void rowpass_fold_pair_kernel(GemmArgs g, int tiles_n, int ntiles) {
  __shared__ __attribute__((aligned(16))) Tile lds;
  __shared__ __attribute__((aligned(16))) double epi_a[128];
  rowpass_block<1, true>(g, tiles_n, ntiles, lds, epi_a);
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// Can this contraction take the specialised kernel?  (full 128-column tiles of the inducing dimension, k-extent a
// multiple of 16 for the forward, even leading dimensions and 16-byte aligned operands, none of the general kernel's extras)
bool gemm_rowpass_eligible(const GemmArgs& g) {
  // [r5] role 1 also as the in-place update C -= op(A) op(B) (alpha = -1, beta = 1, no statistics, any k-extent that is a
  // multiple of 16): launch_gemm_rowpass sets c_sub
  const bool sub = g.role == 1 && g.alpha == -1.0 && g.beta == 1.0 && !g.fs_part && g.b_tri == 0 && g.store_c;
  if (g.nouter != 1 || (!sub && (g.alpha != 1.0 || g.beta != 0.0)) || g.win || g.M_last || g.N_last || g.K_last || g.a_tri) return false;
  if ((g.lda & 1) || (g.ldb & 1) || (g.ldc & 1) || !aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C)) return false;
  if (g.c_src && (!sub || !aligned16(g.c_src))) return false;
  if (g.fs_k && (g.role != 1 || !g.fs_part || g.fs_sq || !aligned16(g.fs_k))) return false;
  if ((g.sA & 1) || (g.sB & 1) || (g.sC & 1) || (g.sSplit & 1)) return false;
  if (g.role == 1)
    return !g.a_kmajor && g.b_kmajor && !g.lower_only && g.ksplit == 1 && !g.kscale && g.b_tri >= 0 && (g.N % BN) == 0 &&
           (g.K % BK) == 0 && (g.K == g.N || sub) && g.K >= BK && g.M >= 1;
  if (g.role == 2)
    return g.a_kmajor && g.b_kmajor && g.kscale && g.b_tri == 0 && g.M == g.N && (g.N % BN) == 0 &&
           g.lda == g.ldb && g.K >= 1;
  return false;
}

void launch_gemm_rowpass(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  g.c_sub = (g.role == 1 && g.alpha == -1.0 && g.beta == 1.0) ? 1 : 0;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int ntiles = g.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
  const int gx = (g.ksplit > 1) ? ntiles * ((g.ksplit + 7) / 8) * 8 : ntiles;
  dim3 grid(gx, 1, g.nbatch);
  if (g.role == 1) {
    // triangular fold: column tiles paired (j, tiles_n - 1 - j) per block (see the kernel)
    const int pair = (g.b_tri > 0 && g.ksplit == 1 && tiles_n >= 2 && (tiles_n & 1) == 0) ? 1 : 0;
    if (pair) {
      grid.x = tiles_m * (tiles_n / 2);
      hipLaunchKernelGGL(rowpass_fold_pair_kernel, grid, dim3(NT), 0, stream, g, tiles_n, (int)grid.x);
    } else {
      hipLaunchKernelGGL((rowpass_gemm_kernel<1>), grid, dim3(NT), 0, stream, g, tiles_n, ntiles);
    }
  } else {
    hipLaunchKernelGGL((rowpass_gemm_kernel<2>), grid, dim3(NT), 0, stream, g, tiles_n, ntiles);
  }
}

static bool rowpass_enabled() {
  static const bool enabled = [] {   // HMOGP_ROWPASS=0 forces the general kernel (A/B measurements)
    const char* e = getenv("HMOGP_ROWPASS");
    return !(e && e[0] == '0');
  }();
  return enabled;
}
bool gemm_rowpass_would_take(const GemmArgs& g) { return rowpass_enabled() && gemm_rowpass_eligible(g); }

int launch_gemm_rowpass_or_general(const GemmArgs& g, hipStream_t stream) {
  const bool enabled = rowpass_enabled();
  if ((g.fs_sq || g.fs_k) && !(enabled && gemm_rowpass_eligible(g)))
    throw HipError{hipErrorInvalidValue, "fs_sq / fs_k are features of the specialised forward kernel only", __FILE__, __LINE__};
  if (enabled && gemm_rowpass_eligible(g)) {
    launch_gemm_rowpass(g, stream);
    return NWN;
  }
  if (g.role == 1 && (g.alpha != 1.0 || g.beta != 0.0)) {   // the general kernel's role-1 epilogue stores op(A) op(B) as it is
    GemmArgs h = g;
    h.role = 0;
    launch_gemm_f64(h, stream);
    return 2;
  }
  launch_gemm_f64(g, stream);
  return 2;
}
