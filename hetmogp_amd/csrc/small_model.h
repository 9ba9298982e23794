// small_model.h -- the fused small-model kernels (small_model.hip): M <= HMOGP_SMALL_M, one block per latent GP.
#pragma once
#include "common.h"
#include "rowpass.h"   // SmallQuadRed

#define HMOGP_SMALL_M 64

struct SmallU {   // inputs / outputs of u_small_kernel (all device pointers; per-latent stride M*M unless noted)
  int M = 0, Q = 0, P = 1, ldz = 0;
  const double* Z = nullptr;       // [M][Q*P]
  const double* var = nullptr;     // [Q]
  const double* ell = nullptr;     // [Q]
  const double* jit = nullptr;     // [Q] jitter added to the factorised copy of K_uu (0, or a forced rung's value)
  const double* mu = nullptr;      // [M][Q]
  const double* Lflat = nullptr;   // [M(M+1)/2][Q]
  double *Kuu = nullptr, *Luu = nullptr, *Kuui = nullptr, *L = nullptr, *S = nullptr, *KiS = nullptr, *KSK = nullptr, *C = nullptr,
         *Ctri = nullptr, *Sqi = nullptr;
  double* a = nullptr;             // [Q][M]
  double* klout = nullptr;         // [Q][KL_BLOCKS][5]
  int* info = nullptr;             // [Q] LAPACK info of the factorisation (always written: 0 = fine)
  long long* stamps = nullptr;     // diagnostics (HMOGP_USMALL_STAMPS): [Q][2][16] s_memtime at the phase boundaries of each block
  int regs = 1;                    // the wave-level chains with their matrix in registers (0: in LDS)
  int stop_after = 0;              // diagnostics (HMOGP_USMALL_STOP): leave block (q, 0) after phase 1..3 -- timing only
  int* flag = nullptr;             // [Q] hand-over flag of S between the two blocks of a latent: set to *seq when S is in HBM
  const double* seq = nullptr;     // evaluation counter in the parameter block (a new value per evaluation: no memset of the flags)
  double* zero = nullptr;          // the statistic bundle, zeroed here by the blocks (q, 1) (the row pass accumulates into it) ...
  long long nzero = 0;             // ... its length (0: leave it alone)
};

struct SmallF {   // finish_small_kernel
  int M = 0, Q = 0, want_qu = 1, want_hz = 1;
  long long per_q = 0, oR = 0;     // bundle: H_q at H + q * per_q (lower triangle), r_q at + oR
  const double* H = nullptr;
  double* Hfull = nullptr;         // the same buffer: receives the mirrored upper triangle
  const double *Kuui = nullptr, *KiS = nullptr, *KSK = nullptr, *Sqi = nullptr, *L = nullptr, *a = nullptr;
  double *G = nullptr, *GSK = nullptr, *dLdS = nullptr, *dKmm = nullptr, *Kr = nullptr;
  double* gL = nullptr;            // [M(M+1)/2][Q]
  double* gmu = nullptr;           // [M][Q]
  double *gL2 = nullptr, *gmu2 = nullptr;   // optional second copies (the engine's D2H staging block: one copy for all results)
  // gradients_X / update_gradients_full of dL_dKmm reduced against K_zz (post.hip kzz_rows_kernel, same arithmetic), by block (q, 1)
  const double *Z = nullptr, *var = nullptr, *ell = nullptr;
  int P = 1, ldz = 0;
  double* rowout = nullptr;        // [Q][M][2 + P] (nullptr: not wanted)
  // The results of the evaluation, gathered by the LAST block of the grid to finish straight into the caller's page-locked host
  // block (no gather launch, no copy node):  [ head of the bundle (n_hg) | KL partials (n_kl) | per-latent tails (Q x n_tail, from
  // bundle + n_hg' + q per_q + oDZ) | rowout (n_row) | extra (n_extra: the q(u) gradients of gmu2 / gL2) | info (Q, as doubles) ]
  double* stage = nullptr;         // device-visible address of the host block (nullptr: no gather)
  const double *g_stats = nullptr, *g_kl = nullptr, *g_extra = nullptr;
  const int* g_info = nullptr;
  long long n_hg = 0, n_kl = 0, n_tail = 0, oDZ = 0, n_row = 0, n_extra = 0, NG = 0;
  int* counter = nullptr;          // blocks finished so far (zero between launches: the last block resets it)
};


struct SmallRows {   // small_fwd_kernel / small_bwd_kernel / small_red_kernel: the row pass of one pool of n rows, 64 rows per block
  int M = 0, Q = 0, P = 1, ldz = 0, hyper = 1, want_z = 1;
  long long n = 0, ldn = 0;        // rows of the pool, row stride of the per-latent row vectors / K^ / P~ workspaces
  const double* X = nullptr;       // [n][P] inputs of the pool's rows
  const double* Z = nullptr;       // [M][Q*P]
  const double *var = nullptr, *ell = nullptr;   // [Q]
  const double* C = nullptr;       // [Q][M][M]
  const double* a = nullptr;       // [Q][M]
  double *Kh = nullptr, *Pt = nullptr;           // [Q][ldn][M]
  double *vp = nullptr, *vc = nullptr, *vpt = nullptr, *vct = nullptr;   // [Q][ldn] (vpt / vct unused without hyper)
  const double *alpha = nullptr, *beta = nullptr, *alpha0 = nullptr, *beta0 = nullptr;   // [Q][ldn] row weights (backward)
  double* slab = nullptr;          // [nblk][Q][M*M + M + M*P] block partials of H_q | r_q | dZ_q
  double* stats = nullptr;         // bundle (H_q at stats + NG + q * per_q, r at + oR, dZ at + oDZ): receives the sums
  long long NG = 0, per_q = 0, oR = 0, oDZ = 0;
};
size_t small_rows_lds_bytes();
void launch_small_fwd(const SmallRows& r, hipStream_t s);
// block partials + their deterministic sum into the bundle (+ the quadrature's partials, when given)
void launch_small_bwd(const SmallRows& r, hipStream_t s, const SmallQuadRed* qr = nullptr);

size_t small_lds_bytes();
void launch_u_small(const SmallU& u, hipStream_t s);
void launch_finish_small(const SmallF& f, hipStream_t s);
