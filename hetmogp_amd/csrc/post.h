// post.h -- launchers of the element-wise / reduction kernels of the replicated M x M algebra (post.hip).
#pragma once
#include "common.h"

void launch_add_diag_copy(const double* src, double* dst, int Q, int M, const double* d_jit, hipStream_t s);
void launch_sub(const double* A, const double* B, double* C, long long n, hipStream_t s);
void launch_tri_fold(const double* C, double* T, int Q, int M, hipStream_t s);  // T = tril(C) + tril(C^T, -1)
void launch_identity(double* A, int Q, int M, hipStream_t s);   // A[q] = I
void launch_strict_d(const double* KiS, double* D, int Q, int M, hipStream_t s);  // D = KiS^T - I  (strict q(f), two-solve form)
// [r6] strict q(f), one-solve form: V[q] = [ L_q^T ; Kuu^-1 S - I ; m_q^T ] (2 M + 1 rows of M, stride sV) for ONE forward row-solve
// against Luu, and its result taken apart: W = Luu^-1 L_q, W2 = Luu^-1 (S Kuu^-1 - I) (both k-major = row-major), w3 = Luu^-1 m
void launch_strict_stack(const double* L, const double* KiS, const double* mu, double* V, long long sV, int Q, int M, hipStream_t s);
void launch_strict_unstack(const double* V, long long sV, double* W, double* W2, double* w3, int Q, int M, hipStream_t s);
void launch_cond_probe(const double* Kuui, const double* var, int Q, int M, double* out, hipStream_t s);  // out[q] = var_q max diag
void launch_transpose_batched(const double* A, long long sA, double* B, long long sB, int Q, int M, hipStream_t s);  // B[q] = A[q]^T
#define KL_BLOCKS 64
// out[(q*KL_BLOCKS + b)*5 + {0..4}] = block partials of sum(Kuui.*S), m^T a, sum log|diag Luu|, sum log|diag L|,
// #inf(Sqi) (Sqi may be nullptr); the host adds the KL_BLOCKS partials in order
void launch_kl_terms(const double* Kuui, const double* S, const double* m_u, const double* a, const double* Luu,
                     const double* L, const double* Sqi, int Q, int M, double* out, hipStream_t s);
void launch_dkmm(const double* G, const double* GSK, const double* Kuui, const double* KSK, const double* Kr, const double* a,
                 double* out, int Q, int M, hipStream_t s);
void launch_dlds(const double* G, const double* Kuui, const double* Sqi, double* out, long long n, hipStream_t s);
void launch_pack_gl(const double* T, double* gL, int Q, int M, hipStream_t s);
void launch_gmu(const double* Kr, const double* a, double* g, int Q, int M, hipStream_t s);
// rowout[q][m][0..2+P) = { sum_j EK, sum_j EK r2, sum_j (EK + EK^T)(z_j - z_m)[p] }
void launch_kzz_rows(const double* dKmm, const double* Z, int ldz, int P, const double* d_var, const double* d_ell, int Q,
                     int M, double* rowout, hipStream_t s);
void launch_qf_combine(const double* p, const double* c, long long ldn, long long N, int Q, int Df, const double* W,
                       const double* kappa, const double* var, double* m, double* v, hipStream_t s);
// natural-gradient step of q(u)
void launch_natgrad_prec(const double* Sqi, const double* dLdS, double gamma, double* out, int Q, int M, bool reversed,
                         hipStream_t s);
void launch_commit_if_ok(const int* info, int Q, const double* src1, double* dst1, long long n1, const double* src2, double* dst2,
                         long long n2, hipStream_t s);
void launch_antitranspose(const double* T, double* L, int Q, int M, hipStream_t s);  // L[i][j] = T[M-1-j][M-1-i]
void launch_gemv_t_batched(const double* A, const double* x, double* y, int Q, int M, hipStream_t s);  // y = A^T x, [Q][M]
void launch_natgrad_theta1(const double* t1, const double* t2, const double* gm, double gamma, double* out, int Q, int M,
                           hipStream_t s);
void launch_pack_tril(const double* T, double* out, int Q, int M, double scale, hipStream_t s);
void launch_scatter_mq(const double* v, double* out, int Q, int M, hipStream_t s);
// device-resident Adadelta (climin recurrence, util.py:327) on one parameter block; phase 0 = momentum move before the
// gradient, phase 1 = update from grad (nullptr = zero gradient); omd = 1 - d as the host computes it
void launch_adadelta(double* x, double* gms, double* sms, double* step, double* pend, const double* grad, double sign, long long n,
                     int phase, double rate, double m, double d, double omd, double o, hipStream_t s);
// the small results of one evaluation gathered into one contiguous block (one D2H copy instead of 3 + Q)
void launch_gather_small(const double* stats, long long n_hg, const double* kl, long long n_kl, long long per_q, long long oDZ,
                         long long n_tail, int Q, const double* rowout, long long n_row, double* dst, hipStream_t s);
