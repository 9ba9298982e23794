// gemm_f64.hip -- FP64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), gfx950 only.
//
// One kernel serves every dense contraction of the svmogp_inf path:
//   forward   P~ = K^ C_q                 (N x M x M, A row-major -> transposing LDS stage)
//   backward  H += K^T diag(beta) K^      (M x M x N, both operands k-major, k-scaled, split over N, lower tiles)
//   M x M     Kuu^-1 S, G = Kuu^-1 H Kuu^-1, trailing Cholesky updates, triangular-inverse merges, ...
//
// Tiling (wave64, 4 waves = one per SIMD): block tile 128 x 128 x 16, wave tile 64 x 64 = 4 x 4 MFMA tiles of
// 16 x 16, 16 independent accumulators per wave (128 VGPRs) so the 64-cycle FP64 MFMA issues back to back.
// Both operands are staged k-major in LDS ([k][row], leading dimension 144 doubles): the MFMA A/B fragment of
// lane l is element [k = l>>4][row = l&15], so a wave reads 4 runs of 16 consecutive doubles; 144*8 B = 288
// dwords == 32 banks (mod 64) puts the two k-rows of each 32-lane half on disjoint banks (ds_read_b64 is
// conflict-free).  LDS is double buffered; the next tile's global loads are issued before the MFMA block.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NTHREADS = 256;
// LDS images of one 128 x 16 operand tile (2304 doubles either way):
//   k-major  [k][144]  for operands stored [k][row]  -> fragment (lane l) = [k = l>>4][row = l&15]
//   row-major [row][18] for operands stored [row][k] -> copied as is (no transposition), fragment = [row = l&15][k = l>>4]
// 144*8 B = 288 dwords == 32 (mod 64) and 18*8 B = 36 dwords: in both images the 32 lanes of a ds_read_b64 half-wave hit
// 64 distinct banks, and the 16-byte staging stores of each 8-lane group hit 32 distinct banks.
constexpr int KM_LD = 144, RM_LD = 18, TILE_DOUBLES = BK * KM_LD;
static_assert(BM * RM_LD == TILE_DOUBLES, "both LDS images have the same size");

struct Tile {
  double a[2][TILE_DOUBLES];
  double b[2][TILE_DOUBLES];
};

// Load 8 consecutive doubles p[0..7] (16-byte vector loads) -- caller guarantees alignment and bounds.
__device__ __forceinline__ void load8_fast(const double* __restrict__ p, double (&v)[8]) {
  const f64x2* q = reinterpret_cast<const f64x2*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f64x2 t = q[i];
    v[2 * i] = t.x;
    v[2 * i + 1] = t.y;
  }
}

// Operand stored [row][k] (k contiguous): thread loads row r = t>>1, 8 k's starting at (t&1)*8.
__device__ __forceinline__ void load_rowmajor(const double* __restrict__ base, int ld, int row0, int nrows, int k0,
                                              int kend, bool fast, double (&v)[8]) {
  const int t = threadIdx.x, r = t >> 1, kh = (t & 1) * 8;
  const double* p = base + (long long)(row0 + r) * ld + (k0 + kh);
  if (fast) {
    load8_fast(p, v);
  } else {
    const bool rok = (row0 + r) < nrows;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = 0.0;
      if (rok && (k0 + kh + i) < kend) v[i] = p[i];
    }
  }
}
__device__ __forceinline__ void store_rowmajor(double* s, const double (&v)[8]) {
  const int t = threadIdx.x, r = t >> 1, kh = (t & 1) * 8;
  f64x2* q = reinterpret_cast<f64x2*>(s + r * RM_LD + kh);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = f64x2{v[2 * i], v[2 * i + 1]};
}

// Operand stored [k][col] (col contiguous): thread owns row k = t>>4 and the four column pairs
// j*32 + (t&15)*2 (j = 0..3): each 16-lane group reads / writes 256 contiguous bytes per vector instruction
// (coalesced global segments, conflict-free ds_write_b128).
__device__ __forceinline__ void load_kmajor(const double* __restrict__ base, int ld, int col0, int ncols, int k0,
                                            int kend, bool fast, double (&v)[8]) {
  const int t = threadIdx.x, k = t >> 4, c2 = (t & 15) * 2;
  const double* p = base + (long long)(k0 + k) * ld + (col0 + c2);
  if (fast) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f64x2 x = *reinterpret_cast<const f64x2*>(p + j * 32);
      v[2 * j] = x.x;
      v[2 * j + 1] = x.y;
    }
  } else {
    const bool kok = (k0 + k) < kend;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v[2 * j + e] = 0.0;
        if (kok && (col0 + c2 + j * 32 + e) < ncols) v[2 * j + e] = p[j * 32 + e];
      }
  }
}
__device__ __forceinline__ void store_kmajor(double* s, const double (&v)[8]) {
  const int t = threadIdx.x, k = t >> 4, c2 = (t & 15) * 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f64x2*>(s + k * KM_LD + c2 + j * 32) = f64x2{v[2 * j], v[2 * j + 1]};
}

// ROLE only names the instantiation (0 = generic M x M algebra, 1 = forward P~ = K^ C, 2 = weighted Gram) so that
// rocprofv3 reports the two row-pass contractions separately from the small replicated GEMMs.
template <bool A_KMAJOR, bool B_KMAJOR, int ROLE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f64_kernel(GemmArgs g, int tiles_n, int ntiles) {
  __shared__ __attribute__((aligned(16))) Tile lds;
  __shared__ __attribute__((aligned(16))) double epi_a[ROLE == 1 ? 128 : 2];  // fused epilogue: vector a of the block's columns

  // ---- which tile / batch / k-range -----------------------------------------------------------------
  // Block b is observed to run on XCD b % 8 (speed only, never correctness).  Without a K split every XCD gets a
  // contiguous range of tiles (the column tiles of one row panel share its A panel through that XCD's L2).  With a
  // K split, all tiles of one K range go to ONE XCD: they stream the same rows of the operand concurrently, so the
  // operand is fetched from HBM once per range instead of once per tile.
  int v = blockIdx.x, split = 0;
  if (g.ksplit > 1) {
    const int xcd = v & 7, idx = v >> 3;
    split = (idx / ntiles) * 8 + xcd;
    v = idx % ntiles;
    if (split >= g.ksplit) return;
  } else if ((ntiles & 7) == 0) {
    const int cpx = ntiles >> 3;
    v = (v & 7) * cpx + (v >> 3);
  }
  int ti, tj;
  if (g.lower_only) {
    ti = (int)((sqrt(8.0 * (double)v + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= v) ++ti;
    while (ti * (ti + 1) / 2 > v) --ti;
    tj = v - ti * (ti + 1) / 2;
  } else {
    ti = v / tiles_n;
    tj = v - ti * tiles_n;
  }
  const int batch = blockIdx.z;
  int M = g.M, N = g.N, K = g.K;
  if (batch == g.nbatch - 1) {
    if (g.M_last > 0) M = g.M_last;
    if (g.N_last > 0) N = g.N_last;
    if (g.K_last > 0) K = g.K_last;
  }
  const int i0 = ti * BM, j0 = tj * BN;
  if (i0 >= M || j0 >= N) return;
  int wlo = 0, whi = K;
  const int* win = g.win ? g.win + (long long)batch * g.win_stride : nullptr;
  double* fs_part = g.fs_part ? g.fs_part + (long long)batch * g.fs_sPart : nullptr;
  if (ROLE == 1 && win) {  // exact-zero window of this row tile (see GemmArgs::win)
    wlo = win[2 * ti], whi = min(K, win[2 * ti + 1]);
    if (j0 >= whi || j0 + BN <= wlo) {  // no consumer ever reads this P~ tile: its statistics are exact zeros
      if (fs_part && threadIdx.x < 128 && i0 + (int)threadIdx.x < M)
        for (int st = 0; st < (g.fs_hyper ? 4 : 2); ++st)
          for (int hf = 0; hf < 2; ++hf)
            fs_part[((long long)st * 2 * tiles_n + 2 * tj + hf) * M + (i0 + threadIdx.x)] = 0.0;
      return;
    }
  }
  if (g.a_tri > 0) whi = min(whi, i0 + BM);       // triangular operands (GemmArgs::a_tri / b_tri): trimmed k-range
  if (g.b_tri < 0) whi = min(whi, j0 + BN);
  if (g.a_tri < 0) wlo = max(wlo, i0);
  if (g.b_tri > 0) wlo = max(wlo, j0);
  wlo = min(wlo, whi);
  if (ROLE == 2 && win) {
    wlo = max(win[2 * ti], win[2 * tj]);
    whi = max(wlo, min(K, min(win[2 * ti + 1], win[2 * tj + 1])));
  }
  const int ksteps = (whi - wlo + BK - 1) / BK;
  const int per = (ksteps + g.ksplit - 1) / g.ksplit;
  const int kbeg = wlo + split * per * BK;
  const int kend = min(whi, kbeg + per * BK);

  const long long ob = blockIdx.y;
  const double* __restrict__ A = g.A + (long long)batch * g.sA + ob * g.oA;
  const double* __restrict__ B = g.B + (long long)batch * g.sB + ob * g.oB;
  const double* __restrict__ S = g.kscale ? g.kscale + (long long)batch * g.sS + ob * g.oS : nullptr;
  double* __restrict__ C = g.C + (long long)batch * g.sC + ob * g.oC + (long long)split * g.sSplit;

  // vector-load eligibility (uniform per block)
  const bool alignA = ((g.lda & 1) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool alignB = ((g.ldb & 1) == 0) && ((((uintptr_t)B) & 15) == 0);
  const bool fullA = alignA && (i0 + BM <= M);
  const bool fullB = alignB && (j0 + BN <= N);

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  // Lower-only products, diagonal tiles.  ROLE 2 (the weighted Gram, 8 of its 36 tiles at M = 1024) spreads the 36
  // sub-tiles on or below the diagonal over all four waves: the two diagonal quadrants keep their 10 lower sub-tiles
  // each (`mma_mode` 1), and the full quadrant (1,0) is shared by its own wave (sub-tile rows 0..1, mode 2) and the
  // wave whose quadrant (0,1) is never read (rows 2..3, mode 3): at most 10 MFMAs per wave and k4-step instead of 16.
  // Elsewhere (ROLE 0) the wave of the strictly-upper quadrant just idles (it still stages and takes the barriers).
  const bool diag2 = ROLE == 2 && g.lower_only && ti == tj;
  const int wm = (diag2 && w == 1) ? 1 : (w >> 1), wn = (diag2 && w == 1) ? 0 : (w & 1);
  const int mma_mode = !diag2 ? 0 : ((w == 0 || w == 3) ? 1 : (w == 2 ? 2 : 3));
  const bool idle_quadrant = g.lower_only && ti == tj && wm == 0 && wn == 1;
  // ROLE 1 accumulates the TRANSPOSED 16 x 16 sub-tiles (operands swapped in the MFMA) with the B fragment's columns
  // permuted, so that lane (lr, lk) register r of acc[a][b] holds P~[wm*64 + a*16 + lr][wn*64 + b*16 + 4*lk + r]:
  // a lane owns 4 adjacent columns of ONE row -> the row statistics of the fused epilogue are in-lane sums plus two
  // shuffle steps, and P~ is stored with 16-byte vectors.
  constexpr bool SWAP = (ROLE == 1);
  const int blr = SWAP ? 4 * (lr & 3) + (lr >> 2) : lr;

  f64x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

  // The loads of a tile are issued one k-step before they are consumed by `stage`: the k-scale of the Gram operand is
  // therefore applied at stage time, not at load time (a multiply at load time makes hipcc wait vmcnt(0) right behind the
  // loads and exposes the whole HBM latency every k-step).
  double ra[8], rb[8], ks = 1.0;
  auto load = [&](int k0) {
    const bool fk = (k0 + BK <= kend);
    if (A_KMAJOR)
      load_kmajor(A, g.lda, i0, M, k0, kend, fullA && fk, ra);
    else
      load_rowmajor(A, g.lda, i0, M, k0, kend, fullA && fk, ra);
    if (B_KMAJOR) {
      load_kmajor(B, g.ldb, j0, N, k0, kend, fullB && fk, rb);
      if (S) {
        const int k = k0 + (t >> 4);
        ks = (k < kend) ? S[k] : 0.0;
      }
    } else {
      load_rowmajor(B, g.ldb, j0, N, k0, kend, fullB && fk, rb);
    }
  };
  auto stage = [&](int buf) {
    if (A_KMAJOR)
      store_kmajor(lds.a[buf], ra);
    else
      store_rowmajor(lds.a[buf], ra);
    if (B_KMAJOR) {
      if (S) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] *= ks;
      }
      store_kmajor(lds.b[buf], rb);
    } else {
      store_rowmajor(lds.b[buf], rb);
    }
  };

  // one k-step of MFMAs over the sub-tiles of this wave: MODE 0 all 16, 1 the 10 with b <= a, 2 rows a < 2, 3 rows a >= 2
  auto mma = [&](int buf, auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const double* pa = A_KMAJOR ? &lds.a[buf][(kk * 4 + lk) * KM_LD + wm * 64 + lr] : &lds.a[buf][(wm * 64 + lr) * RM_LD + kk * 4 + lk];
      const double* pb = B_KMAJOR ? &lds.b[buf][(kk * 4 + lk) * KM_LD + wn * 64 + blr] : &lds.b[buf][(wn * 64 + blr) * RM_LD + kk * 4 + lk];
      double fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!(MODE == 2 && i >= 2) && !(MODE == 3 && i < 2)) fa[i] = pa[i * 16 * (A_KMAJOR ? 1 : RM_LD)];
        fb[i] = pb[i * 16 * (B_KMAJOR ? 1 : RM_LD)];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if ((MODE == 1 && b > a) || (MODE == 2 && a >= 2) || (MODE == 3 && a < 2)) continue;
          acc[a][b] = SWAP ? __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
  };

  // stage-first software pipeline (see gemm_rowpass.hip): step k + 1 goes from registers to the other buffer before the MFMA
  // block of step k, the registers are refilled with step k + 2 at once
  int cur = 0;
  if (kbeg < kend) {
    load(kbeg);
    stage(0);
    if (kbeg + BK < kend) load(kbeg + BK);
  }
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if (k0 + BK < kend) stage(cur ^ 1);
    if (k0 + 2 * BK < kend) load(k0 + 2 * BK);
    if (ROLE == 2 && mma_mode != 0) {
      if (mma_mode == 1) mma(cur, std::integral_constant<int, 1>{});
      else if (mma_mode == 2) mma(cur, std::integral_constant<int, 2>{});
      else mma(cur, std::integral_constant<int, 3>{});
    } else if (!idle_quadrant) {
      mma(cur, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    cur ^= 1;
  }


  // ---- epilogue: D fragment of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg -----------
  const double alpha = g.alpha, beta = (g.ksplit > 1) ? 0.0 : g.beta;
  if (ROLE == 1 && fs_part) {
    // Fused row statistics (see GemmArgs).  Lane (lr, lk) owns row rl = wm*64 + a*16 + lr and the column quads
    // wn*64 + b*16 + 4*lk + (0..3), b = 0..3, of slice a: the statistics need the SAME elements of the K^ tile, so every
    // lane re-reads exactly its own 16 values per slice (L2/MALL-warm) -- no cross-lane exchange.  They are fetched with
    // LDS-DMA loads (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16*l) into a wave-private double buffer,
    // slice a+1 in flight while slice a is consumed, using no VGPRs (the accumulators fill the register file) and no
    // block barriers.  Rows beyond the matrix are clamped (never written); ragged / unaligned column tiles take plain
    // clamped loads.
    const int P = g.fs_P;
    const bool hyper = g.fs_hyper != 0;
    const bool dma = ((g.lda & 1) == 0) && ((((uintptr_t)A) & 15) == 0) && (j0 + BN <= N);
    double* flat = &lds.a[0][0];                  // Tile = 4 * TILE_DOUBLES contiguous doubles
    double* stage = flat + w * 2048;              // [2 buffers][8 chunks (b, h)][64 lanes][2]
    double* xs = flat + 4 * 2048;                 // [4][128] inputs of the block's rows / lengthscale (dimension-major)
    double* zs = xs + 4 * 128;                    // [4][128] inducing inputs of the block's columns / lengthscale
    static_assert(4 * 2048 + 2 * 4 * 128 <= 4 * TILE_DOUBLES, "epilogue scratch fits the tile buffers");
    const int colq = j0 + wn * 64 + 4 * lk;       // + b*16 (+ 2h)
    auto issue = [&](int a) {
      const int grow = min(i0 + wm * 64 + a * 16 + lr, M - 1);
      const double* src = A + (long long)grow * g.lda + colq;
      double* dst = stage + (a & 1) * 1024;
      if (dma) {
#pragma unroll
        for (int c = 0; c < 8; ++c)               // chunk c = 2*b + h
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (c >> 1) * 16 + (c & 1) * 2),
                                           (void __attribute__((address_space(3)))*)(dst + c * 128), 16, 0, 0);
      } else {                                    // ragged / unaligned tile: same slots, plain clamped loads
        const double* row = A + (long long)grow * g.lda;
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
          const int col = colq + (c >> 1) * 16 + (c & 1) * 2;
          *reinterpret_cast<f64x2*>(dst + c * 128 + 2 * lane) = f64x2{row[min(col, N - 1)], row[min(col + 1, N - 1)]};
        }
      }
    };
    issue(0);
    {
      const double inv_l = 1.0 / g.fs_ell[batch];
      for (int e = t; e < 4 * 128; e += NTHREADS) {
        const int p = e >> 7, rr = e & 127;
        xs[e] = (hyper && i0 + rr < M && p < P) ? g.fs_x[(long long)(i0 + rr) * P + p] * inv_l : 0.0;
        zs[e] = (hyper && j0 + rr < N && p < P) ? g.fs_z[(long long)batch * g.fs_sZ + (long long)(j0 + rr) * g.fs_ldz + p] * inv_l : 0.0;
      }
      if (t < 128) epi_a[t] = (j0 + t < N) ? g.fs_a[(long long)batch * g.fs_sA + j0 + t] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // slice a has landed (and the previous partial stores)
      if (a < 3) issue(a + 1);
      const int rl = wm * 64 + a * 16 + lr;
      double sp = 0.0, sc = 0.0, spt = 0.0, sct = 0.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int cl = wn * 64 + b * 16 + 4 * lk;  // this lane's 4 adjacent columns of sub-tile b (within the tile)
        const double* kp = stage + (a & 1) * 1024 + (2 * b) * 128 + 2 * lane;
        const f64x2 k01 = *reinterpret_cast<const f64x2*>(kp), k23 = *reinterpret_cast<const f64x2*>(kp + 128);
        const double kv[4] = {k01.x, k01.y, k23.x, k23.y};
        const f64x2 a01 = *reinterpret_cast<const f64x2*>(epi_a + cl), a23 = *reinterpret_cast<const f64x2*>(epi_a + cl + 2);
        const double av[4] = {a01.x, a01.y, a23.x, a23.y};
        double r2[4] = {0.0, 0.0, 0.0, 0.0};
        if (hyper) {
#pragma unroll 1
          for (int p = 0; p < P; ++p) {              // uniform trip count (P <= 4); rolled: bounded register pressure
            const f64x2 z01 = *reinterpret_cast<const f64x2*>(zs + p * 128 + cl), z23 = *reinterpret_cast<const f64x2*>(zs + p * 128 + cl + 2);
            const double zz[4] = {z01.x, z01.y, z23.x, z23.y};
            const double xp = xs[p * 128 + rl];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double d = xp - zz[r];
              r2[r] += d * d;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double pv = acc[a][b][r];
          sp += kv[r] * av[r];
          sc += pv * kv[r];
          if (hyper) {
            const double wv = kv[r] * r2[r];
            spt += wv * av[r];
            sct += pv * wv;
          }
        }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {           // the four lanes (lk) that share this row
        sp += __shfl_xor(sp, o, 64);
        sc += __shfl_xor(sc, o, 64);
        if (hyper) {
          spt += __shfl_xor(spt, o, 64);
          sct += __shfl_xor(sct, o, 64);
        }
      }
      if (lk == 0 && i0 + rl < M) {                  // partial of column half (tj, wn): [stat][2*tiles_n][M]
        double* o = fs_part + ((long long)(2 * tj + wn)) * M + (i0 + rl);
        const long long ss = 2LL * tiles_n * M;
        o[0] = sp, o[ss] = sc;
        if (hyper) o[2 * ss] = spt, o[3 * ss] = sct;
      }
    }
    if (!g.store_c) return;
  }
  if (idle_quadrant) return;  // the strictly-upper quadrant is never read (mirrored from the lower one)
  if (SWAP) {  // lane (lr, lk): row wm*64 + a*16 + lr, columns wn*64 + b*16 + 4*lk + (0..3)
    const bool vecC = ((g.ldc & 1) == 0) && ((((uintptr_t)C) & 15) == 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = i0 + wm * 64 + a * 16 + lr;
      if (row >= M) continue;
      double* crow = C + (long long)row * g.ldc;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = j0 + wn * 64 + b * 16 + 4 * lk;
        if (vecC && col + 3 < N && beta == 0.0) {
          *reinterpret_cast<f64x2*>(crow + col) = f64x2{alpha * acc[a][b][0], alpha * acc[a][b][1]};
          *reinterpret_cast<f64x2*>(crow + col + 2) = f64x2{alpha * acc[a][b][2], alpha * acc[a][b][3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < N) {
              double val = alpha * acc[a][b][r];
              if (beta != 0.0) val += beta * crow[col + r];
              crow[col + r] = val;
            }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if ((mma_mode == 2 && a >= 2) || (mma_mode == 3 && a < 2)) continue;  // the other wave of the shared quadrant
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + wm * 64 + a * 16 + 4 * r + lk;
      if (row >= M) continue;
      double* crow = C + (long long)row * g.ldc;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = j0 + wn * 64 + b * 16 + lr;
        if (col < N) {
          double val = alpha * acc[a][b][r];
          if (beta != 0.0) val += beta * crow[col];
          crow[col] = val;
        }
      }
    }
  }
}

}  // namespace

namespace {
__global__ void combine_parts_kernel(const double* __restrict__ part, int tiles, long long n, double* __restrict__ p,
                                     double* __restrict__ c, double* __restrict__ pt, double* __restrict__ ct,
                                     long long sPart, long long ldn) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  part += (long long)blockIdx.y * sPart;
  const long long ob = (long long)blockIdx.y * ldn;
  double* outs[4] = {p ? p + ob : nullptr, c ? c + ob : nullptr, pt ? pt + ob : nullptr, ct ? ct + ob : nullptr};
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    if (!outs[st]) continue;
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += part[((long long)st * tiles + t) * n + i];
    outs[st][i] = s;
  }
}
}  // namespace

void launch_combine_parts(const double* part, int tiles, long long n, double* p, double* c, double* pt, double* ct,
                          hipStream_t s, int nb, long long sPart, long long ldn) {
  if (n <= 0) return;
  hipLaunchKernelGGL(combine_parts_kernel, dim3((unsigned)((n + 255) / 256), nb), dim3(256), 0, s, part, tiles, n, p, c, pt,
                     ct, sPart, ldn);
}

void launch_gemm_f64(const GemmArgs& g, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || g.nbatch <= 0 || g.nouter <= 0) return;
  if (g.kscale && !g.b_kmajor) throw HipError{hipErrorInvalidValue, "k-scale needs a k-major B operand", __FILE__, __LINE__};
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int ntiles = g.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
  // fewer than two 128 x 128 blocks per CU: one wave per SIMD reaches half the MFMA rate -> 64 x 64 tiles (gemm_small.hip)
  static const bool small_enabled = [] {
    const char* e = getenv("HMOGP_SMALL_GEMM");
    return !(e && e[0] == '0');
  }();
  if (small_enabled && (long long)tiles_m * tiles_n * g.nouter * g.nbatch < 512 && gemm_small_eligible(g)) {
    launch_gemm_small(g, stream);
    return;
  }
  const int gx = (g.ksplit > 1) ? ntiles * ((g.ksplit + 7) / 8) * 8 : ntiles;
  dim3 grid(gx, g.nouter, g.nbatch), block(NTHREADS);
  if (g.role == 1 && !g.a_kmajor && g.b_kmajor)
    hipLaunchKernelGGL((gemm_f64_kernel<false, true, 1>), grid, block, 0, stream, g, tiles_n, ntiles);
  else if (g.role == 2 && g.a_kmajor && g.b_kmajor)
    hipLaunchKernelGGL((gemm_f64_kernel<true, true, 2>), grid, block, 0, stream, g, tiles_n, ntiles);
  else if (g.a_kmajor && g.b_kmajor)
    hipLaunchKernelGGL((gemm_f64_kernel<true, true, 0>), grid, block, 0, stream, g, tiles_n, ntiles);
  else if (g.a_kmajor && !g.b_kmajor)
    hipLaunchKernelGGL((gemm_f64_kernel<true, false, 0>), grid, block, 0, stream, g, tiles_n, ntiles);
  else if (!g.a_kmajor && g.b_kmajor)
    hipLaunchKernelGGL((gemm_f64_kernel<false, true, 0>), grid, block, 0, stream, g, tiles_n, ntiles);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<false, false, 0>), grid, block, 0, stream, g, tiles_n, ntiles);
}
