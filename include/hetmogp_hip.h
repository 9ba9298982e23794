/*
 * hetmogp_hip.h -- C ABI of the MI355X-native SVI engine for heterogeneous multi-output GPs.
 *
 * Drop-in boundary for ONE hot path of pmorenoz/HetMOGP: the ELBO-and-gradient evaluation that the
 * reference performs in SVMOGP.parameters_changed()  (hetmogp/svmogp.py:85-166), i.e.
 *   SVMOGPInf.inference()            hetmogp/svmogp_inf.py:23-109
 *     util.latent_funs_cov           hetmogp/util.py:181-200      (K_uu, jitchol, K_uu^-1)
 *     SVMOGPInf.calculate_q_f        hetmogp/svmogp_inf.py:186-225 (q(f_d) mean / variance)
 *     HetLikelihood.var_exp(_derivatives)  hetmogp/het_likelihood.py:101-131 -> likelihoods/<name>.py
 *     SVMOGPInf.calculate_KL         hetmogp/svmogp_inf.py:227-250
 *     SVMOGPInf.calculate_gradients  hetmogp/svmogp_inf.py:111-183
 *   + the parameter-gradient assembly hetmogp/svmogp.py:101-166, util.py:228-231,248-255.
 *
 * The reference has no FFI (it is pure Python over GPy); the binding a maintainer would add is the
 * ctypes stub shown in INTEGRATION.md.  Why the boundary sits at the OUTER level (parameters in,
 * parameter gradients out) and not at SVMOGPInf.inference(): the inner protocol returns dL_dKmn as
 * Q*Df dense M x N matrices (24.6 GB at N=200k, M=1024) for Python to reduce.
 *
 * Conventions: every array is C-contiguous float64 (int32 / int64 where stated) in HOST memory owned by
 * the caller; the engine copies in / out and keeps data and workspaces resident in HBM between calls.
 * One handle = one host thread = one HIP device.  Every function returns 0 on success or a negative
 * HMOGP_E_* code; hmogp_last_error() gives the message.  The library has no CPU fallback: without a
 * HIP device hmogp_create fails with HMOGP_E_NO_DEVICE.
 */
#ifndef HETMOGP_HIP_H
#define HETMOGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMOGP_ABI_VERSION 8

/* likelihood ids (class names of /root/reference/likelihoods/<name>.py) */
enum {
  HMOGP_LIK_GAUSSIAN = 0,    /* gaussian.py     param = sigma (default 0.5)          dim_f = 1   */
  HMOGP_LIK_BERNOULLI = 1,   /* bernoulli.py                                          dim_f = 1   */
  HMOGP_LIK_HETGAUSSIAN = 2, /* hetgaussian.py                                        dim_f = 2   */
  HMOGP_LIK_CATEGORICAL = 3, /* categorical.py  param = K (labels 1..K)               dim_f = K-1 */
  HMOGP_LIK_POISSON = 4,     /* poisson.py                                            dim_f = 1   */
  HMOGP_LIK_EXPONENTIAL = 5, /* exponential.py                                        dim_f = 1   */
  HMOGP_LIK_GAMMA = 6,       /* gamma.py                                              dim_f = 2   */
  HMOGP_LIK_BETA = 7         /* beta.py                                               dim_f = 2   */
};

/* error codes; the Python facade maps them onto the reference's exception types */
enum {
  HMOGP_OK = 0,
  HMOGP_E_INVALID = -1,     /* bad argument                                                        */
  HMOGP_E_NO_DEVICE = -2,   /* no HIP device / HIP runtime error                                    */
  HMOGP_E_NOT_PD = -3,      /* LinAlgError("not positive definite, even with jitter.") / non-positive
                               diagonal -- GPy jitchol, reached from util.py:198                    */
  HMOGP_E_SQI_UNSTABLE = -4,/* ValueError("Sqi: Cholesky representation unstable") svmogp_inf.py:126 */
  HMOGP_E_STATE = -5,       /* call order violated (finish without begin, data not set, ...)        */
  HMOGP_E_COMM = -6         /* RCCL not loadable / a collective failed (hmogp_comm_*); ABI version 4  */
};

/* output flags (hmogp_outputs.flags) */
#define HMOGP_FLAG_V_NEGATIVE 1u /* some v_fd < 0: the reference prints 'v negative!' (svmogp_inf.py:221) */
#define HMOGP_FLAG_ILL_CONDITIONED 2u /* ABI v6: hmogp_outputs.cond_est of some latent is beyond what this engine's mode keeps within
   the north-star's element-wise 1e-5 of the reference: > 5e2 without HMOGP_CFG_STRICT_QF (cond(K_uu) ~ 1e4 and more: take the
   strict mode), > 1e6 with it (cond beyond the ~2e7 that GPy's jitter rung 0 leaves: the reference's own numbers move by more than 1e-5 with the last bit of its
   covariance there).  The reference has no such diagnostic (GPy's jitchol only reports a failed factorisation).              */

/* Reference quirks (hmogp_config.quirks).  A set bit reproduces the reference's behaviour, a cleared
 * bit gives the mathematically exact quantity.  HMOGP_QUIRKS_REFERENCE (all set) is what parity is measured against;
 * HMOGP_QUIRKS_EXACT (0) makes every returned gradient the true gradient of the returned ELBO (finite-difference
 * tested), which is what a quasi-Newton optimiser needs.                                                          */
#define HMOGP_QUIRK_GAMMA_BETA_PI 1u   /* Q1: Gamma / Beta var_exp and derivatives are 1/pi x the 2-D quadrature
                                          (gamma.py:110,139-141; beta.py:113,142-144); exact: the quadrature itself */
#define HMOGP_QUIRK_CATEGORICAL_DM 2u  /* Q2: Categorical d var_exp / dm = onehot(y)[d] - 1, independent of f
                                          (categorical.py:102-113); exact: E[onehot(y)[d] - softmax_d(f)]            */
#define HMOGP_QUIRK_STALE_W 4u         /* Q3: chain factors W0 / kappa0 (construction-time copies, svmogp.py:98,141,143,
                                          156) are honoured; exact: hmogp_params.W0 / kappa0 are ignored (= W / kappa) */
#define HMOGP_QUIRK_W_DIAG 8u          /* Q4: dW from the K_ff diagonal = W * sum(gv)   (util.py:230); exact:
                                          2 * W * variance_q * sum(gv)                                               */
#define HMOGP_QUIRK_KAPPA_DIAG 16u     /* Q5: dkappa = sum(gv)  (util.py:231); exact: variance_q * sum(gv)            */
#define HMOGP_QUIRKS_REFERENCE 31u
#define HMOGP_QUIRKS_EXACT 0u

/* gradient-group mask (hmogp_params.group_mask): which parameter groups receive a gradient.  Mirrors the
 * VEM gating of svmogp.py:104-110,131-137,145-151,160-166 and Z.is_fixed (:153,158).                    */
#define HMOGP_GROUP_QU 1u     /* m_u, L_u            (zeroed in stochastic M-steps)                 */
#define HMOGP_GROUP_HYPER 2u  /* variance, lengthscale, W, kappa (zeroed in stochastic E-steps)      */
#define HMOGP_GROUP_Z 4u      /* inducing inputs     (zeroed in stochastic E-steps or when fixed)    */
#define HMOGP_GROUP_ALL 7u

typedef struct hmogp_engine* hmogp_handle;

/* Static description of the model: SVMOGP.__init__ (svmogp.py:17-79) + HetLikelihood.generate_metadata
 * (het_likelihood.py:24-44).                                                                          */
typedef struct {
  int32_t abi_version;      /* HMOGP_ABI_VERSION                                                     */
  int32_t T;                /* number of likelihoods / tasks  = len(Y)                               */
  int32_t Q;                /* number of latent GPs u_q       = len(kern_list)                       */
  int32_t M;                /* inducing points per latent GP  = Z.shape[0]                           */
  int32_t P;                /* input dimension (Xdim)                                                */
  int32_t Df;               /* number of latent parameter functions f_d = sum_t dim_f(t)             */
  const int32_t* lik_id;    /* [T]  HMOGP_LIK_*                                                      */
  const double* lik_param;  /* [T]  sigma (Gaussian) | K (Categorical) | ignored                     */
  const int32_t* f_index;   /* [Df] task owning function d      (Y_metadata['function_index'])        */
  const int32_t* d_index;   /* [Df] column of d inside its task (Y_metadata['d_index'])               */
  int32_t device;           /* HIP device ordinal                                                    */
  int64_t chunk_rows;       /* rows streamed per pass, across tasks (0 = default: up to 2^20, less when
                               2 * Q * rows * M * 8 bytes of N x M workspace would exceed ~64 GB)     */
  uint32_t flags;           /* HMOGP_CFG_*                                                           */
  uint32_t quirks;          /* HMOGP_QUIRK_* mask; HMOGP_QUIRKS_REFERENCE reproduces the reference             */
} hmogp_config;
/* Engine limits (hmogp_create fails with HMOGP_E_INVALID beyond them): 1 <= P <= 4 input dimensions (the distance
 * kernels are instantiated per P); Q <= 8 latent GPs; dim_f <= 8 functions per likelihood (Categorical K <= 9);
 * exact-zero windows: M <= 8192.  T, M, Df and the row counts are bounded by device memory only.                 */

/* hmogp_config.flags */
#define HMOGP_CFG_STRICT_QF 8u /* ABI v6, opt-in: STRICT q(f).  The reference forms q(f_d) and the row side of its gradients through
   A = K_fu K_uu^-1 obtained by triangular solves against Luu (dpotrs, svmogp_inf.py:214-218; A^T alpha, A^T diag(beta) A and
   A (S K_uu^-1 - I) at :144-161); the default path uses the algebraically equal explicit C_q = K_uu^-1 S K_uu^-1 - K_uu^-1, which
   differs from that by ~cond(K_uu) * 2^-53.  Once GPy's jitter ladder is taken (cond ~ 1e7) the default path's g_W / g_kappa / g_Z
   are 1e-4 .. 1e-3 away from the reference's; with this flag the engine follows the reference's forms (two blocked triangular
   solves against Luu, the Gram of A, ...) and stays within 1e-5 element-wise there (tests/test_gpu_ladder.py).  Regular kernels
   only (no fused small-model path), excludes HMOGP_CFG_EXACT_ZERO_WINDOWS.  bench.py's `value` never uses it.
   (ABI v8) Cost at the headline size: 1.6x the default step for an evaluation that asks for the q(u) gradients only (136 vs 84 ms),
   1.8x for a full-gradient evaluation (213 vs 119.5 ms; 2.5x in ABI v7) while the condition estimate of K_uu is <= 1e6 -- GPy's whole
   jitter-ladder regime --, 2.2x beyond (263 ms).  Up to 1e6 an evaluation takes the ONE-SOLVE form: only the forward substitution
   X = K_fu Luu^-T touches the n x M side, the backward half of dpotrs sits in M x M factors (A m = X (Luu^-1 m), A L_q =
   X (Luu^-1 L_q), rowsum(A .* K_fu) = rowsum(X .* X), A^T diag(b) A = Luu^-T (X^T diag(b) X) Luu^-1, A (S K_uu^-1 - I) =
   X (Luu^-1 (S K_uu^-1 - I))); beyond it the literal two-solve form (the two are equally far from the reference's
   operations up to there, and the one-solve form drifts out of the reference's own sensitivity beyond).                       */
#define HMOGP_CFG_NO_SMALL_PATH 4u /* ABI v5: keep the regular kernels and three streams also for small models (M <= 64 would
                                    * otherwise take the fused small-model kernels, M <= 128 with <= 65536 rows one stream): A/B
                                    * comparisons of the two paths inside one process (tests)                                */
#define HMOGP_CFG_CACHE_KUU 2u /* opt-in: reuse K_uu, its Cholesky factor and inverse while (Z, variance, lengthscale,
   forced rungs) are bit-identical to the previous evaluation's (VEM / SVI E-steps only move q(u)).  Off by default and
   never used by bench.py, whose steps re-evaluate identical parameters.                                          */
#define HMOGP_CFG_EXACT_ZERO_WINDOWS 1u /* opt-in: skip the parts of K_uf = k(X,Z) that are EXACTLY 0.0 in float64
   (exp underflow beyond ~38.6 lengthscales).  For spatially sorted inputs K_uf is banded and the row pass only touches
   the band; every skipped term is a product with an exact zero, so ELBO and gradients are unchanged (tested equal to
   the dense path); unsorted inputs fall back to dense ranges on the device.  Default off: the dense path is what
   bench.py reports as `value`.                                                                                  */

/* All model parameters of one evaluation: the arrays paramz hands to parameters_changed().             */
typedef struct {
  const double* Z;            /* [M, Q*P]  block q = inducing inputs of u_q (svmogp.py:52)            */
  const double* m_u;          /* [M, Q]    q_u_means                  } both NULL: use the device-resident q(u)  */
  const double* L_flat;       /* [M(M+1)/2, Q] q_u_chols, GPy row-major tril packing } (hmogp_qu_load / _adadelta) */
  const double* variance;     /* [Q]  RBF variance of k_q                                             */
  const double* lengthscale;  /* [Q]  RBF lengthscale of k_q                                          */
  const double* W;            /* [Q, Df] live coregionalisation weights  B_list[q].W                  */
  const double* kappa;        /* [Q, Df] live kappa                      B_list[q].kappa              */
  const double* W0;           /* [Q, Df] or NULL (= W): construction-time W_list used for the chain
                                 factors at svmogp.py:98,141,143,156             */
  const double* kappa0;       /* [Q, Df] or NULL (= kappa)                                            */
  const double* batch_scale;  /* [T] or NULL (= 1): N_all[t] / N_batch[t]  (svmogp.py:89-90)          */
  const int64_t* row_begin;   /* [T] or NULL (= 0):   first row of task t used in this evaluation     */
  const int64_t* row_end;     /* [T] or NULL (= N_t): one past the last row (contiguous minibatch
                                 slices of util.py:52-72; also the row shard of this rank)            */
  const int32_t* forced_rung; /* [Q] or NULL: -2 = run GPy's jitter ladder, -1 = no jitter,
                                 k>=0 = jitter mean(diag)*1e-6*10^k (compare CPU/GPU at equal rung)   */
  uint32_t group_mask;        /* HMOGP_GROUP_*                                                        */
  uint32_t eval_flags;        /* HMOGP_EVAL_* of THIS evaluation (0 = none; (ABI v8) unknown bits are refused
                                 with HMOGP_E_INVALID: zero the struct before filling it)   (ABI version 6) */
} hmogp_params;
#define HMOGP_EVAL_STRICT_QF 1u /* run this evaluation in the strict q(f) mode (see HMOGP_CFG_STRICT_QF) whatever the engine was
   created with: lets a caller re-evaluate, and go on evaluating, in that mode once hmogp_outputs.flags reported
   HMOGP_FLAG_ILL_CONDITIONED -- the facade's strict_qf="auto".  Its extra workspaces are allocated at the first such call. */
#define HMOGP_EVAL_NO_G_L 2u /* (ABI v7) the gradient of q(u)'s Cholesky factor is not wanted from this evaluation: dL/dS L and its packing
                             * are skipped (outputs.g_L_u, if given, is zero-filled).  A natural-gradient E-step consumes dL/dS and dL/dm
                             * only (hmogp_natgrad_step / hmogp_qu_natgrad*).  hmogp_qu_adadelta phase 1 refuses to follow such an
                             * evaluation (HMOGP_E_STATE).  Ignored by the fused small-model path (M <= 64), which forms it anyway. */

typedef struct {
  double* elbo;          /* [1]   log_marginal (svmogp_inf.py:88)                                     */
  double* g_m_u;         /* [M, Q]                                                                    */
  double* g_L_u;         /* [M(M+1)/2, Q]                                                             */
  double* g_variance;    /* [Q]                                                                       */
  double* g_lengthscale; /* [Q]                                                                       */
  double* g_W;           /* [Q, Df]                                                                   */
  double* g_kappa;       /* [Q, Df]                                                                   */
  double* g_Z;           /* [M, Q*P]                                                                  */
  double* dL_dS;         /* [Q, M, M] or NULL: dL/dS_q (svmogp_inf.py:169), for natural gradients     */
  int32_t* rung;         /* [Q] jitter rung taken per latent (-1 = none)                              */
  uint32_t* flags;       /* [1] HMOGP_FLAG_*                                                          */
  double* kl;            /* [Q] or NULL: KL(q(u_q) || p(u_q)) per latent (calculate_KL, svmogp_inf.py:227-250);
                            elbo = (scaled data term) - sum_q kl[q]                      (ABI version 3) */
  double* cond_est;      /* [Q] or NULL: variance_q * max_i (K_uu^-1)_ii of the (jittered) prior covariance of latent q: a lower
                            bound of its condition number, 30-150x below it for RBF kernels          (ABI version 6) */
} hmogp_outputs;

/* ---- life cycle ------------------------------------------------------------------------------------ */
int hmogp_create(const hmogp_config* cfg, hmogp_handle* out);
void hmogp_destroy(hmogp_handle h);
const char* hmogp_last_error(hmogp_handle h); /* h may be NULL: error of the last failed hmogp_create    */
int hmogp_abi_version(void);

/* Upload (replace) the full data of task t: X [N, P], Y [N] (Xmulti_all[t], Ymulti_all[t]).            */
int hmogp_set_task_data(hmogp_handle h, int32_t t, const double* X, const double* Y, int64_t N);

/* ---- the hot path ----------------------------------------------------------------------------------- */
/* One full evaluation = parameters_changed(): ELBO + all parameter gradients (single device).           */
int hmogp_elbo_grad(hmogp_handle h, const hmogp_params* p, hmogp_outputs* out);
/* Small models (M <= 64, see HMOGP_CFG_NO_SMALL_PATH): the evaluation is a fixed launch sequence on one stream; the second
 * hmogp_elbo_grad with the same gradient gates / row ranges / forced rungs is captured into a hipGraph and every later one is a
 * replay (one hipGraphLaunch instead of ~35 API calls; the parameter VALUES travel through a page-locked image the graph's
 * upload node reads).  hmogp_graph_stats reports how many graphs were captured and how many evaluations were replays
 * (diagnostics / tests; environment HMOGP_SMALL_GRAPH=0 disables the mechanism).  ABI v5.                                  */
int hmogp_graph_stats(hmogp_handle h, int64_t* captures, int64_t* replays);

/* The same, split at the one exchange point of the path for row-sharded multi-GPU runs:
 *   begin  : upload parameters, replicated M x M pre-algebra, row pass over [row_begin,row_end) of every
 *            task -> additive statistic bundle left in device memory (synchronous at return);
 *   (caller sum-all-reduces the bundle in place across ranks, e.g. torch.distributed over RCCL)
 *   finish : replicated M x M post-processing of the bundle -> outputs.                                 */
int hmogp_step_begin(hmogp_handle h, const hmogp_params* p);
int hmogp_stats_buffer(hmogp_handle h, void** device_ptr, int64_t* count); /* float64 words            */
int hmogp_step_finish(hmogp_handle h, hmogp_outputs* out);
/* Host-staged access to the bundle (tests).  Between begin and finish H_q holds its LOWER triangle only (the row
 * pass fills nothing else; finish mirrors it), so sums of bundles of different row shards stay valid bundles.     */
int hmogp_stats_read(hmogp_handle h, double* host /* [count] */);
int hmogp_stats_write(hmogp_handle h, const double* host /* [count] */);
/* Wire format of the bundle = what actually travels in the exchange step: H_q is symmetric, so only its lower
 * triangle is sent -- [head 2+Df | per q: tril(H_q) row-major packed, M(M+1)/2 | r (M) | dZ (M*P) | sa | sl | swk (Df)]
 * = 12.7 MB instead of 25.2 MB at M = 1024, Q = 3.  Usage between begin and finish:
 *   hmogp_wire_pack(h);  all-reduce(sum, float64) the `count` words at `device_ptr` in place;  hmogp_wire_unpack(h);
 * pack / unpack are synchronous at return.  hmogp_wire_read / _write: host-staged variant (CPU backends, tests).   */
int hmogp_wire_buffer(hmogp_handle h, void** device_ptr, int64_t* count);
int hmogp_wire_pack(hmogp_handle h);
int hmogp_wire_unpack(hmogp_handle h);
int hmogp_wire_read(hmogp_handle h, double* host /* [count] */);
int hmogp_wire_write(hmogp_handle h, const double* host /* [count] */);

/* ---- the exchange step inside the library (ABI version 4) --------------------------------------------------------
 * The reference is single-process (no distributed code at all); the contract is SURVEY.md 8(e): rows are
 * sharded over one process per GPU and the additive bundle is sum-all-reduced ONCE per step.  With a communicator
 * attached the library does that itself: wire pack -> ncclAllReduce(sum, float64, in place) -> wire unpack, all enqueued
 * on the engine's own HIP stream between the row pass and the replicated post-processing -- no host synchronisation, no
 * second library's stream.  librccl is resolved at run time (dlopen by soname: in a process that already holds one,
 * e.g. PyTorch's, that copy is used), so single-GPU users do not need it.
 *   hmogp_comm_available   1 if librccl could be loaded, else 0 (no handle needed; never fails)
 *   hmogp_comm_unique_id   ncclGetUniqueId: ONE rank calls it and distributes the HMOGP_COMM_ID_BYTES bytes out of band
 *                          (torch.distributed broadcast, MPI, a file ...)
 *   hmogp_comm_init        ncclCommInitRank on the engine's device; collective: every rank calls it with the same id
 *   hmogp_comm_destroy     ncclCommDestroy (also done by hmogp_destroy)
 *   hmogp_comm_info        nranks / rank of the attached communicator (0 / -1 without one)
 * hmogp_elbo_grad_sharded (ABI v5) IS the row-sharded step: begin on this rank's rows -> exchange -> finish, one call, one
 * final host synchronisation; COLLECTIVE -- every rank of the communicator must call it.  hmogp_elbo_grad itself never
 * communicates, also with a communicator attached (ABI v4 keyed the collective on the communicator's presence: a debug
 * call on one rank would then block its peers).  hmogp_step_exchange is the middle part for callers that keep the
 * three-call form.  Failure semantics: a rank whose row pass fails before it could contribute ABORTS the communicator
 * (ncclCommAbort) so that the peers' collective ends with HMOGP_E_COMM instead of blocking; while a collective is in
 * flight the final wait polls ncclCommGetAsyncError and a deadline (environment HMOGP_COMM_TIMEOUT_S, default 600,
 * 0 = none) and aborts the communicator on either.  After an abort hmogp_comm_info reports 0 ranks.  The exchange is
 * category [8] of hmogp_last_timings.  A communicator of ONE rank runs the same three launches (used by the tests).   */
#define HMOGP_COMM_ID_BYTES 128
int hmogp_comm_available(void);
int hmogp_comm_unique_id(void* id /* [HMOGP_COMM_ID_BYTES] */);
int hmogp_comm_init(hmogp_handle h, int32_t nranks, int32_t rank, const void* id /* [HMOGP_COMM_ID_BYTES] */);
int hmogp_comm_destroy(hmogp_handle h);
int hmogp_comm_info(hmogp_handle h, int32_t* nranks, int32_t* rank);
int hmogp_step_exchange(hmogp_handle h);
int hmogp_elbo_grad_sharded(hmogp_handle h, const hmogp_params* p, hmogp_outputs* out);

/* ---- posterior / prediction (consumers: svmogp.py:238-251, 280-306) --------------------------------- */
/* woodbury_vector[q] = Kuu^-1 m_q  [Q, M];  woodbury_inv[q] = Kuu^-1 - Kuu^-1 S_q Kuu^-1  [Q, M, M]
 * of the parameters of the last evaluation.                                                             */
int hmogp_posterior_u(hmogp_handle h, double* woodbury_vector, double* woodbury_inv);
/* q(f_d) at new inputs for all functions: m [Nnew, Df], v [Nnew, Df] (= predictive_new's (mu, |var|)
 * before the abs; svmogp_inf.py:186-225 with X := Xnew), using the parameters of the last evaluation.   */
int hmogp_predict_f(hmogp_handle h, const double* Xnew, int64_t Nnew, double* m, double* v);

/* ---- natural-gradient update of q(u) (named in the north-star; the reference has no such step) ---------- */
/* From the gradients of the LAST finished evaluation (its group_mask must include HMOGP_GROUP_QU):
 *   S_q^-1 <- S_q^-1 - 2 gamma dL/dS_q ;  S_q^-1 m_q <- S_q^-1 m_q + gamma (dL/dm_q - 2 dL/dS_q m_q)
 * returns the new m_u [M, Q] and L_flat [M(M+1)/2, Q] (Cholesky of the new S_q, GPy packing).  HMOGP_E_NOT_PD if
 * the step leaves the positive-definite cone (nothing is modified: retry with a smaller gamma).  A successful step
 * invalidates posterior_u / predict_f until the next evaluation.                                                       */
int hmogp_natgrad_step(hmogp_handle h, double gamma, double* m_u_new, double* L_flat_new);
/* The same step applied IN PLACE to the device-resident q(u) of hmogp_qu_load (ABI v5): nothing but two info words crosses
 * PCIe; the next evaluation (m_u = L_flat = NULL) sees the updated q(u).  HMOGP_E_NOT_PD leaves the resident q(u) and the
 * gradients of the last evaluation untouched (the step only wrote scratch), so the caller can retry at once with a smaller
 * gamma.  After a successful step posterior_u / predict_f / another step need a new evaluation (HMOGP_E_STATE otherwise).  */
int hmogp_qu_natgrad(hmogp_handle h, double gamma);
/* (ABI v7) The same in-place step WITHOUT the host synchronisation.  Everything is enqueued, the commit into the resident q(u) is
 * decided on the device (a step that leaves the positive-definite cone writes nothing), and the call returns at once: the caller
 * can enqueue the next evaluation straight away -- its parameter upload, row staging and K_uf construction then run beside this
 * step's factorisation chain instead of behind a host round trip (0.3 ms per SVI iteration at M = 1024, Q = 3).
 * hmogp_qu_natgrad_status waits for the pending step (if any) and sets *taken = 1 if it was committed, 0 if it was refused (q(u)
 * unchanged; the gradients it was computed from are gone once a new evaluation has run: take a smaller step from the new ones).
 * At most one step may be pending (HMOGP_E_STATE); the synchronous entry points resolve a pending step first.              */
int hmogp_qu_natgrad_async(hmogp_handle h, double gamma);
int hmogp_qu_natgrad_status(hmogp_handle h, int32_t* taken);

/* ---- device-resident q(u) and its Adadelta state: the SVI loop without moving 2 x 12.6 MB per iteration ------ */
/* The reference's stochastic driver (util.py:321-329) runs climin.Adadelta over the flat optimiser vector, 98 % of which
 * is q(u) (m_u, L_u: 1.58 M numbers at M = 1024, Q = 3); its gradient is svmogp.py:188-199 (stochastic_grad).  These
 * entry points keep q(u), its gradient and the three Adadelta accumulators in HBM; the caller runs the same recurrence
 * on the few remaining parameters (Z, variance, W ...) on the host.  The device recurrence performs the same IEEE
 * operations in the same order as the host one, so the iterates are bit-identical.
 *   hmogp_qu_load      upload q(u) (zeroes the accumulators); evaluations with params.m_u = params.L_flat = NULL use it,
 *                      outputs.g_m_u / g_L_u may then be NULL (nothing is copied back)
 *   hmogp_qu_adadelta  phase 0: q(u) -= step2 of the previous iteration; step1 = momentum * step; q(u) -= step1
 *                               (before the gradient evaluation)
 *                      phase 1: gms = d gms + (1-d) g^2; step2 = sqrt(sms+o)/sqrt(gms+o) g rate (kept pending);
 *                               step = step1 + step2; sms = d sms + (1-d) step^2       with g = -dELBO/dq(u) of the last
 *                               evaluation, or 0 if its group_mask excluded HMOGP_GROUP_QU (the M-steps of svmogp.py:196)
 *   hmogp_qu_read      download q(u) as of the LAST EVALUATION (the second half-step of the last update still pending):
 *                      what the reference's model object holds between iterations -- climin updates its own vector, the
 *                      model is only written by stochastic_grad (svmogp.py:188)                                       */
int hmogp_qu_load(hmogp_handle h, const double* m_u, const double* L_flat);
int hmogp_qu_read(hmogp_handle h, double* m_u, double* L_flat);
int hmogp_qu_adadelta(hmogp_handle h, int32_t phase, double step_rate, double momentum, double decay,
                      double one_minus_decay, double offset);

/* ---- timing --------------------------------------------------------------------------------------- */
/* Milliseconds the kernels of the last evaluation spent, measured with HIP events on the engine's stream:
 * out[0] whole evaluation, [1] K_uf construction (rbf_cross_cov; + window kernels in the opt-in mode), [2] forward
 * N x M x M contraction incl. its fused row-statistics epilogue (ONE kernel per launch, all latents of a task chunk),
 * [3] combine of the row-statistic partials, [4] quadrature, [5] weighted Gram contraction (ONE kernel per launch),
 * [6] column statistics + slab reductions, [7] replicated M x M algebra, [8] the in-library exchange step (pack + RCCL
 * all-reduce + unpack; 0 without a communicator), [9] the triangular solves of the strict q(f) mode (trsm_panel_kernel /
 * trsm_diag_kernel + their update GEMMs; [2] then holds the mode's products T = A L_q and P~ = A (S K_uu^-1 - I) only),
 * [10] the strict mode's row-statistic kernels (0 in the default mode).  launches[i] = kernel launches behind out[i].  Both
 * arrays hold HMOGP_NTIMINGS entries (8 before ABI version 4, 9 before ABI version 8).                                */
#define HMOGP_NTIMINGS 11
int hmogp_last_timings(hmogp_handle h, double* out_ms /* [HMOGP_NTIMINGS] */, int64_t* launches /* [HMOGP_NTIMINGS] or NULL */);

/* ---- inner protocol, debug / parity mode (small N only) ------------------------------------------------ */
/* The raw gradient dictionary SVMOGPInf.inference returns (svmogp_inf.py:107, built at :130-171) for the LAST finished
 * evaluation, which must have used group_mask = HMOGP_GROUP_ALL and streamed all its rows in one pool (N_total <=
 * chunk_rows): dL_dKmm [Q, M, M] (:166-171); dL_dKmn = for q, for d: [M, N_t(d)] row-major, concatenated (:157-161);
 * dL_dKdiag = for q, for d: [N_t(d)], concatenated (:164).  N_t(d) = rows of the task owning function d in that
 * evaluation.  Any pointer may be NULL.  This is 8*Q*Df*M*N bytes -- the reason the product boundary is the outer
 * protocol; it exists so that the device's dL_dKmm / dL_dKmn can be compared with the reference's with nothing in
 * between.                                                                                                       */
int hmogp_debug_raw_grads(hmogp_handle h, double* dL_dKmm, double* dL_dKmn, double* dL_dKdiag);

/* ---- building blocks, exposed for parity tests ("inner protocol" at small sizes) ---------------------- */
/* K = variance * exp(-0.5 * |x - z|^2 / lengthscale^2), GPy RBF.K(X, Z) semantics (util.py:161,197).      */
int hmogp_rbf_cross_cov(int32_t device, const double* X, int64_t N, const double* Z, int32_t M, int32_t P,
                        double variance, double lengthscale, double* K /* [N, M] */);
/* The same with the rounding order selectable: exact = 1 is hmogp_rbf_cross_cov (GPy's r = sqrt(clip(r2))/l, used
 * for K_uu); exact = 0 is the variant the row pass uses for K_uf (clip(r2) * (1/l^2): no sqrt / divide per element,
 * <= 2 ulp of the exponent away).                                                                                 */
int hmogp_rbf_cross_cov_ex(int32_t device, const double* X, int64_t N, const double* Z, int32_t M, int32_t P,
                           double variance, double lengthscale, int32_t exact, double* K /* [N, M] */);
/* Batched lower Cholesky with GPy's jitter ladder + inverse: A [Q,M,M] -> L, Ainv; rung[q] as above.     */
int hmogp_jitchol_inv(int32_t device, const double* A, int32_t Q, int32_t M, const int32_t* forced_rung,
                      double* L, double* Ainv, int32_t* rung);
/* (L L^T)^-1 from lower factors (GPy dpotri): L [Q,M,M] -> Sinv [Q,M,M].                                 */
int hmogp_potri(int32_t device, const double* L, int32_t Q, int32_t M, double* Sinv);
/* out [n, M] = dpotrs(L, B^T)^T = B (L L^T)^-1 for the n rows of B [n, M], L [M, M] lower (GPy util.linalg.dpotrs as
 * svmogp_inf.py:214 calls it): two blocked triangular solves, true substitution inside 32-column diagonal blocks -- the
 * building block of HMOGP_CFG_STRICT_QF.  ABI v6.                                                                      */
int hmogp_potrs_rows(int32_t device, const double* L, int32_t M, const double* B, int64_t n, double* out);
/* C = alpha * op(A) op(B) + beta * C on the FP64-MFMA GEMM; transA/transB in {0,1}; row-major.           */
int hmogp_gemm_f64(int32_t device, int32_t transA, int32_t transB, int32_t M, int32_t N, int32_t K,
                   double alpha, const double* A, int32_t lda, const double* B, int32_t ldb, double beta,
                   double* C, int32_t ldc);
/* Variational expectations of one likelihood: y [N], m,v [N, dim_f] -> ve [N], dm, dv [N, dim_f].         */
int hmogp_var_exp(int32_t device, int32_t lik_id, double lik_param, int64_t N, const double* y,
                  const double* m, const double* v, double* ve, double* dm, double* dv);

/* The same under a quirk mask (HMOGP_QUIRK_GAMMA_BETA_PI, HMOGP_QUIRK_CATEGORICAL_DM).                            */
int hmogp_var_exp_ex(int32_t device, int32_t lik_id, double lik_param, uint32_t quirks, int64_t N, const double* y,
                     const double* m, const double* v, double* ve, double* dm, double* dv);

/* Predictive mean / variance of y under q(f) = N(m, diag v): the reference's `<likelihood>.predictive(m, v)`
 * (e.g. bernoulli.py:113-128, gamma.py:196-238, categorical.py:224-269), consumed by HetLikelihood.predictive
 * (het_likelihood.py:133-148).  m, v [N, dim_f] -> mean, var [N, dim_p] (dim_p = K-1 for Categorical, else 1).
 * gh_T: Gauss-Hermite order, 20 (fresh reference instance), 10 (instance whose var_exp ran first: GPy caches the
 * first rule), 0 = the reference's default for a fresh instance.                                              */
int hmogp_predictive(int32_t device, int32_t lik_id, double lik_param, int32_t gh_T, int64_t N, const double* m,
                     const double* v, double* mean, double* var);

/* Monte-Carlo log predictive density per test row (the inner part of `<likelihood>.log_predictive`, e.g.
 * bernoulli.py:130-144): log_pred[n] = -log(S) + logsumexp_s log p(y_n | f_s), f_s ~ N(m_n, diag v_n), S = num_samples,
 * counter-based generator seeded by `seed` (reproducible; a different stream than NumPy's).  The reference then returns
 * (1/S) * sum_n log_pred[n] and HetLikelihood.negative_log_predictive (het_likelihood.py:150-164) negates the sum over
 * tasks -- done by the caller.  Defined for Gaussian, Bernoulli, HetGaussian, Poisson, Exponential, Categorical.   */
int hmogp_log_predictive(int32_t device, int32_t lik_id, double lik_param, int64_t N, int32_t num_samples, uint64_t seed,
                         const double* y, const double* m, const double* v, double* log_pred);

/* Data generation on the device: one draw Y[n] ~ p(y | F[n, :]) per row with the link functions and clips of the
 * reference's `<likelihood>.samples` (het_likelihood.py:72-83 -> e.g. gamma.py:43-50, categorical.py:65-75; labels of
 * Categorical are 1..K).  Counter-based generator keyed by (seed, row): reproducible, but a different stream than
 * NumPy's -- only the distribution is comparable with the reference.                                             */
int hmogp_sample(int32_t device, int32_t lik_id, double lik_param, int64_t N, uint64_t seed, const double* F /* [N, dim_f] */,
                 double* Y /* [N] */);

/* Micro-benchmark of the two row-pass contractions on synthetic operands resident in HBM (tools/bench_gemm.py):
 * role 1: forward  P~[n,M] = K^[n,M] C[M,M];  role 2: weighted Gram  H[M,M] (lower tiles) = K^T diag(beta) K^ incl. the
 * slab reduction; roles 3 / 4: role 1 with the fused row-statistics epilogue, with / without the P~ store; role 5: K_uf
 * construction alone (rbf_kernel<1, false>, 3 latents batched: 3 * 8 * n * M bytes written per launch); role 6
 * (diagnostic): role 2 over ALL tiles instead of the lower ones, same kernel, no slab reduction (2 n M^2 flops).
 * Returns the average milliseconds per launch over `iters` launches (HIP events).                              */
int hmogp_bench_contraction(int32_t device, int32_t role, int64_t n, int32_t M, int32_t iters, double* avg_ms);

/* Page-locked host memory (hipHostMalloc / hipHostFree).  Optional: parameter and gradient arrays that live in such
 * memory are transferred by DMA without the driver's staging copy (the 12.6 MB L_flat / g_L_u at M = 1024, Q = 3 cost
 * about 1 ms per evaluation from pageable memory).  Returns NULL without a HIP device or on failure.  No reference
 * equivalent.                                                                                                       */
void* hmogp_host_alloc(uint64_t bytes);
void hmogp_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* HETMOGP_HIP_H */
