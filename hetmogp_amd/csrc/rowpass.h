// rowpass.h -- launchers of the row-streaming kernels (rowpass.hip).
#pragma once
#include "common.h"

#define HMOGP_MAXJ 8  // max latent functions per task (Categorical K <= 9)
#define HMOGP_MAXQ 8  // max latent GPs
// scalar statistics one quad block emits: [0] sum ve, [1] #(v<0), [2+2q] sa_q, [3+2q] sl_q, then J x sgv, then Q x J x swk
#define HMOGP_MAXSCAL (2 + 2 * HMOGP_MAXQ + HMOGP_MAXJ + HMOGP_MAXQ * HMOGP_MAXJ)

struct QuadArgs {
  int lik = 0;
  double lik_param = 0.0;
  int dimf = 1, Q = 1;
  long long N = 0;                 // rows of this chunk
  long long off = 0;               // added to the row index of the [Q][ldn] vectors below (0: the pointers are already offset)
  const double* y = nullptr;       // [N]
  const double* yaux = nullptr;    // [N] gammaln(y+1) (Poisson) or nullptr
  const double* p = nullptr;       // [Q][ldn]  K^ a
  const double* c = nullptr;       // [Q][ldn]  rowsum(P~ .* K^)
  const double* pt = nullptr;      // [Q][ldn]  r2-weighted twins (nullptr when no hyper-gradients are wanted)
  const double* ct = nullptr;
  long long ldn = 0;
  double w[HMOGP_MAXQ][HMOGP_MAXJ];    // live W[q][d(j)]
  double w0[HMOGP_MAXQ][HMOGP_MAXJ];   // construction-time W (quirk Q3)
  double kap[HMOGP_MAXQ][HMOGP_MAXJ];  // live kappa
  double var[HMOGP_MAXQ];              // RBF variances
  double scale = 1.0;                  // batch_scale[t]
  // [r4] the same five inputs read from DEVICE memory instead (all non-null together; the by-value copies above are then ignored):
  // what lets a captured hipGraph of the evaluation be replayed with new hyper-parameters.  Wd / W0d / kapd: [Q][Df] (row q,
  // column d0 + j), vard: [Q], scaled: one value
  const double *Wd = nullptr, *W0d = nullptr, *kapd = nullptr, *vard = nullptr, *scaled = nullptr;
  int Df = 0, d0 = 0;
  unsigned quirks = 0x1fu;             // HMOGP_QUIRK_* (default: reproduce the reference)
  double* alpha = nullptr;             // [Q][ldn] outputs: row weights of the backward pass
  double* beta = nullptr;
  double* alpha0 = nullptr;
  double* beta0 = nullptr;
  double* partials = nullptr;          // [nblocks][nscal]
  double* out_mu = nullptr;            // optional [N][dimf]: q(f) mean / variance (prediction, parity tests)
  double* out_v = nullptr;
  double* out_gm = nullptr;            // optional [N][dimf]: scaled d ve / d m, d ve / d v (inner-protocol debug export)
  double* out_gv = nullptr;
  // strict q(f) (HMOGP_CFG_STRICT_QF): p / c above are the solve-based forms (q(f) of svmogp_inf.py:212-218); the block scalars
  // sa / swk take these explicit-inverse forms instead (K^ a, rowsum(P~ .* K^): the reference's gradient code, :157-161)
  const double* pg = nullptr;          // [Q][ldn] or nullptr (= p)
  const double* cg = nullptr;          // [Q][ldn] or nullptr (= c)
};

// strict q(f): the row statistics of the solve-based forms (rowpass.hip: strict_rowstats_kernel)
struct StrictRows {
  int phase = 0, M = 0, Q = 1, P = 1, ldz = 0;
  long long n = 0, ldn = 0, sK = 0, sZ = 0;
  const double *Kh = nullptr, *Ah = nullptr, *Tt = nullptr, *Pt = nullptr;   // [Q][sK] row-major n x M
  const double* mu = nullptr;     // [M][Q] q_u_means                     (two-solve form: Ah holds A = K^ Kuu^-1)
  const double* w3 = nullptr;     // [Q][M] Luu^-1 m_q  ([r6] one-solve form: Ah holds X = K^ Luu^-T; nullptr = two-solve form)
  const double* a = nullptr;      // [Q][M] Kuu^-1 m
  const double *X = nullptr, *Z = nullptr, *ell = nullptr;
  double *p = nullptr, *c = nullptr, *pg = nullptr, *cg = nullptr, *pt = nullptr, *ct = nullptr;   // [Q][ldn]
  const double* t2 = nullptr;     // [Q][ldn] phase 0: rowsum(T .* T) already formed by the product's epilogue (Tt is not read)
};
void launch_strict_rowstats(const StrictRows& a, hipStream_t s);
// one <= 32-column diagonal block of the blocked triangular solves V L^T = B (dir 0) / A L = V (dir 1), in place, batched over Q
// [u0, u1): columns of the same 128-column block that receive the block's right-looking update inside the launch (needs
// trsm_diag_can_fuse; u0 == u1: none)
void launch_trsm_diag(int dir, double* V, long long sV, const double* L, long long sL, int M, int j0, int nb, long long n, int Q,
                      hipStream_t s, int u0 = 0, int u1 = 0, const double* Vsrc = nullptr);
bool trsm_diag_can_fuse(const double* V, long long sV, int M);
// [r6] one 128-column block of the same solves as ONE launch: long-K update + in-tile substitution (trsm_panel.hip)
struct TrsmPanelArgs {
  double* V = nullptr;             // [Q][sV]: n x M row-major (ldv), solved in place
  const double* Vsrc = nullptr;    // optional: this block's right-hand sides are read from here (first touch of the forward solve)
  long long sV = 0;
  int ldv = 0;
  const double* Lsym = nullptr;    // [Q][sL]: the factor mirrored into a full symmetric image (launch_mirror_lower), ldl
  long long sL = 0;
  int ldl = 0;
  const double* rdiag = nullptr;   // [Q][sR]: 1 / Luu[j][j]  (launch_rdiag)
  long long sR = 0;
  long long n = 0;
  int M = 0, j0 = 0, Q = 1;
  // row statistics of the solved tile in the epilogue of the LAST direction a caller runs (nullptr: none):
  //   sp += tile . st_vec[columns],   sk += rowsum(tile .* st_K)  -- or rowsum(tile .* tile) when st_K is null
  const double* st_K = nullptr;    // same layout as V
  const double* st_vec = nullptr;  // element (column j, batch q) at st_vec[q * st_vecB + j * st_vecS]
  long long st_vecB = 0, st_vecS = 1;
  double* st_part = nullptr;       // [Q][st_sPart]: [2 statistics][4 wave columns][st_ld]
  long long st_sPart = 0, st_ld = 0;
};
bool trsm_panel_eligible(const TrsmPanelArgs& g);
void launch_trsm_panel(int dir, const TrsmPanelArgs& g, hipStream_t s);
void launch_rdiag(const double* L, long long sL, int M, int Q, double* out, long long sR, hipStream_t s);
void launch_trsm_stats_combine(const double* part, long long sPart, long long ld, int nparts, long long n, int Q, const double* t2,
                               double* p, double* c, long long ldn, hipStream_t s);

// [r4] every segment (task x row range) of a pool in one launch -- small models only: the weights are read from device memory
#define HMOGP_QUAD_MULTI 8
struct QuadSeg {
  int lik = 0, dimf = 1, d0 = 0, t = 0;  // likelihood id, functions, first function column, task (index of its batch scale)
  double lik_param = 0.0;
  long long N = 0, off = 0;              // rows, first row within the pool's row vectors
  const double* y = nullptr;
  const double* yaux = nullptr;
  unsigned blk0 = 0;                     // first block of the segment        } filled by launch_quad_multi
  long long part0 = 0;                   // its first word in `partials`      }
};
struct QuadMulti {
  int nseg = 0, Q = 1, Df = 0;
  long long ldn = 0;
  const double *p = nullptr, *c = nullptr, *pt = nullptr, *ct = nullptr;
  const double *Wd = nullptr, *W0d = nullptr, *kapd = nullptr, *vard = nullptr, *scale_base = nullptr;
  unsigned quirks = 0x1fu;
  double *alpha = nullptr, *beta = nullptr, *alpha0 = nullptr, *beta0 = nullptr, *partials = nullptr;
  QuadSeg seg[HMOGP_QUAD_MULTI];
};

// per-latent strides of the batched (grid.z = latent) row kernels
struct RbfBatch {
  int nq = 1;
  const double* var = nullptr;  // [nq] device
  const double* ell = nullptr;  // [nq] device
  long long sZ = 0, sK = 0, sWin = 0;
  long long sX = 0;             // per-latent offset of the row inputs (K_uu: the rows are the latent's own inducing inputs)
};
struct ColBatch {
  int nq = 1;
  long long sK = 0, sA = 0, sV = 0, sZ = 0, sPart = 0, sWin = 0;
};

// the quadrature's block partials of every segment of the pool, summed into the bundle by small_red_kernel's extra plane (small models) or by
// reduce_rows_multi_kernel (regular path)
struct SmallQuadRed {
  int nseg = 0;
  struct {
    const double* part = nullptr;  // [nrows][nscal]
    long long nrows = 0;
    int nscal = 0;
    const long long* off = nullptr;   // [nscal] slot -> bundle offset (distinct within a segment)
  } s[8];
};

long long quad_blocks(int lik, long long N);
// [r5] is there an instantiation of quad_multi_kernel for exactly this set of likelihoods (other than the all-inclusive one, which
// runs one wave per SIMD)?
bool quad_multi_specialised(const QuadMulti& m);
// the block partials of every segment of a quad_multi launch -> bundle, segment after segment per slot (the order and the per-
// segment sums of launch_reduce_rows called once per segment: same bits)
void launch_reduce_rows_multi(const SmallQuadRed& qr, double* dst, hipStream_t s);
void launch_quad(const QuadArgs& a, hipStream_t s);
void launch_quad_multi(const QuadMulti& m, hipStream_t s);   // fills blk0 / part0 of the segments; partials laid out segment by segment
void launch_var_exp(int lik, int J, double param, long long N, const double* y, const double* m, const double* v, double* ve,
                    double* dm, double* dv, hipStream_t s, unsigned quirks = 0x1fu);
// K[n][m] = var * exp(-r2/2); X rows have stride ldx, Z rows stride ldz (block q of the M x Q*P inducing array)
// predictive mean / variance of y: m, v [N][J] -> mean, var [N][Jp]; T = Gauss-Hermite order (10 or 20)
void launch_predictive(int lik, int J, int Jp, double param, int T, long long N, const double* m, const double* v,
                       double* mean, double* var, hipStream_t s);
// out[n] = -log S + logsumexp_s log p(y_n | f_s), f_s ~ N(m_n, diag v_n): Monte-Carlo log predictive density per row
void launch_log_predictive(int lik, int J, double param, long long N, int S, unsigned long long seed, const double* y,
                           const double* m, const double* v, double* out, hipStream_t s);
// Y[n] ~ p(y | F[n]): the reference's `<likelihood>.samples`, one draw per row, counter-based generator
void launch_sample(int lik, int J, double param, long long N, unsigned long long seed, const double* F, double* Y,
                   hipStream_t s);
void launch_rbf(const double* X, int ldx, long long N, int P, const double* Z, int ldz, int M, double var, double ell,
                double* K, bool same, hipStream_t s, const int* rowwin = nullptr, bool exact = true,
                const RbfBatch* batch = nullptr);
void launch_reduce_slabs_lower(const double* slabs, int nslabs, int M, double* dst, bool accumulate, hipStream_t s, int nb = 1,
                               long long sSlabs = 0, long long sDst = 0);
// exact-zero windows of K^ = k(X, Z) for one (row chunk, latent): rowwin [tiles][2] column range per 128-row tile,
// colwin [ncb][2] row range per 128-column block, hit [tiles][ncb] scratch (see rowpass.hip)
void launch_windows(const double* X, long long N, int P, const double* Z, int ldz, int M, double ell, int* rowwin, int* colwin,
                    unsigned char* hit, hipStream_t s);
void launch_colstats(const double* Kh, const double* Pt, const double* a, const double* alpha, const double* alpha0,
                     const double* beta0, const double* X, int P, const double* Z, int ldz, long long N, int M, int rows,
                     bool want_z, double* partials, hipStream_t s, const int* colwin = nullptr, const ColBatch* batch = nullptr,
                     int max_blocks = 0, const double* Ar = nullptr, const double* ell = nullptr);
// partial slab per row split: [ r (M) | dZ (M*P) | s2 (M) ], s2[m] = sum_n E_nm |x_n - z_m|^2 / l^2 when `ell` ([nq], device) is given
void launch_sum_cols(const double* v, int Q, int M, double* dst, long long sDst, hipStream_t s);   // dst[q * sDst] += sum_m v[q][m]
void launch_reduce_rows(const double* partials, long long nrows, int len, const long long* off, double* dst, bool accumulate,
                        hipStream_t s);
void launch_reduce_slabs(const double* slabs, int nslabs, long long stride, long long len, double* dst, bool accumulate,
                         hipStream_t s, int nb = 1, long long sSlabs = 0, long long sDst = 0);
void launch_mirror_lower(double* A, int Q, int M, long long stride, hipStream_t s);
// statistic bundle <-> wire format (lower triangles of H_q only); dir 0 = pack, 1 = unpack (lower triangle only)
void launch_wire_copy(double* bundle, double* wire, long long NG, int Q, int M, long long per_q, int dir, hipStream_t s);
// inner-protocol debug export: out[m][n] = a[m] gm[n][j] + 2 w gv[n][j] Pt[n][m]   (svmogp_inf.py:157-161), out is [M][N]
void launch_raw_kmn(const double* a, const double* gm, const double* gv, int J, int j, double w, const double* Pt, int M,
                    long long N, double* out, hipStream_t s);
void launch_gammaln1p(const double* y, double* out, long long N, hipStream_t s);
