// engine_impl.h -- internal declarations shared by the host-side translation units of the engine (engine.hip: orchestration of one
// evaluation; engine_rows.hip: the replicated M x M chain and the row pass; engine_linalg.hip: jitchol ladder and the blocked
// triangular solves; engine_comm.hip: the RCCL exchange step; engine_graph.hip: hipGraph capture of the small path;
// engine_optim.hip: device-resident q(u) optimisers, natural gradient, prediction, the raw-gradient debug export; abi.hip: the
// C ABI of include/hetmogp_hip.h).  [r6] engine.hip was ONE 2 720-line translation unit: split without a
// behaviour change -- every function body is the one it was.  Not part of the public interface.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <chrono>
#include <thread>

#include <dlfcn.h>
#include <hip/hip_runtime.h>
// RCCL is a RUN-TIME dependency only (dlopen, see RcclApi): the few types and constants of its C API this file needs are
// declared here, so the single-GPU library builds on a ROCm install without the rccl development headers.  Where the header
// exists the local declarations are checked against it.
extern "C" {
typedef struct ncclComm* ncclComm_t;
}
namespace hm_nccl {
struct UniqueId { char internal[128]; };
enum : int { Success = 0, InProgress = 7, DataDouble = 8, OpSum = 0 };
using GetUniqueId = int (*)(UniqueId*);
using CommInitRank = int (*)(ncclComm_t*, int, UniqueId, int);
using CommDestroy = int (*)(ncclComm_t);
using CommAbort = int (*)(ncclComm_t);
using CommGetAsyncError = int (*)(ncclComm_t, int*);
using AllReduce = int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
using GetErrorString = const char* (*)(int);
}  // namespace hm_nccl
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == sizeof(hm_nccl::UniqueId), "ncclUniqueId size");
static_assert((int)ncclSuccess == hm_nccl::Success && (int)ncclInProgress == hm_nccl::InProgress &&
              (int)ncclDouble == hm_nccl::DataDouble && (int)ncclSum == hm_nccl::OpSum, "RCCL enum values");
#endif
#endif

#include "../../include/hetmogp_hip.h"
#include "common.h"
#include "post.h"
#include "rowpass.h"
#include "small_model.h"

namespace hmogp_detail {


struct EngineError {
  int code;
  std::string msg;
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr, o.bytes = 0; }
  ~DevBuf() { release(); }
  bool owned = true;
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  void view(void* ptr, size_t b) {   // non-owning window into another allocation
    release();
    p = ptr, bytes = b, owned = false;
  }
  void ensure(size_t b, bool zero = false) {
    if (b <= bytes) return;
    release();
    HIP_TRY(hipMalloc(&p, b));
    bytes = b;
    if (zero) {  // the engine's stream is non-blocking: make the null-stream memset visible before any kernel uses p
      HIP_TRY(hipMemset(p, 0, b));
      HIP_TRY(hipDeviceSynchronize());
    }
  }
  double* d() const { return static_cast<double*>(p); }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

constexpr int FWD_PARTS = GEMM_MAX_FWD_PARTS;  // buffer sizing: partials of the fused row statistics per 128-column tile

enum { CAT_TOTAL = 0, CAT_RBF, CAT_FWD, CAT_ROWSTATS, CAT_QUAD, CAT_GRAM, CAT_COLSTATS, CAT_MM, CAT_EXCHANGE, CAT_TRSM, CAT_STRICT_STATS, NCAT };
static_assert(NCAT == HMOGP_NTIMINGS, "hmogp_last_timings layout");

struct Task {
  long long N = 0;
  DevBuf X, Y, Yaux;
  int lik = 0, dimf = 1, d0 = 0;
  double param = 0.0;
  DevBuf offsets;  // device: quad scalar slot -> bundle offset
  int nscal = 0;
};

int lik_dimf(int lik, double param);

// Row ranges per weighted-Gram launch: a multiple of 8 (one range per XCD at a time), each >= 32 k-steps of 16 rows,
// enough blocks (lower tiles x ranges) for >= 8 rounds over the 256 CUs, at most KS_MAX slabs.
constexpr int KS_MAX = 256;  // most row ranges (slabs) of the weighted Gram
// rows per slab of the column statistics: 256, or 32 for short passes (one thread owns two columns and walks the rows of its
// slab one after the other: at M = 50 a 256-row slab is 25 threads x 256 dependent steps, 110 us for 3000 rows)
inline long long col_split(long long n) { return n <= 16384 ? 32 : 256; }
int gram_ksplit(long long n, int M);

// ------------------------------------------------------------------------------------ batched jitchol + inverse
// Luu <- chol(Kuu + jitter I) with GPy's ladder (GPy.util.linalg.jitchol): plain factorisation first, then
// jitter = mean(diag) * 1e-6 * 10^k, k = 0..4.  diag(K_uu) == variance for the RBF, so mean(diag) = variance.
// rung_io[q]: in  -2 = search, -1 / k = forced;  out = rung taken.
// Two halves so that the caller can enqueue other (independent) work between the asynchronous part and the one host
// synchronisation of the path (the ladder decision).
struct JitcholState {
  std::vector<double> jit;
  std::vector<int> forced, info_own;
  bool complete = false;       // every panel has been enqueued
  int* info = nullptr;         // where the device's info lands: PAGE-LOCKED memory when the caller provides it (a D2H copy
                               // into pageable memory blocks the host until the stream has drained, which would serialise
                               // everything the caller wants to enqueue behind the factorisation)
};
// part 0: set-up + the first `head` panels; part 1: the remaining panels + the info read-back; part -1: everything
constexpr int JIT_HEAD_PANELS = 12;
void jitchol_enqueue(const double* Kuu, double* Luu, int Q, int M, const double* diag_mean, int* rung_io, int* d_info,
                     double* d_jit, double* dscr, hipStream_t st, JitcholState& js, int part = -1);
void jitchol_resolve(const double* Kuu, double* Luu, int Q, int M, const double* diag_mean, int* rung_io, int* d_info,
                     double* d_jit, double* dscr, hipStream_t st, JitcholState& js);
void jitchol_batched(const double* Kuu, double* Luu, int Q, int M, const double* diag_mean, int* rung_io, int* d_info,
                     double* d_jit, double* dscr, hipStream_t st);

// V <- V Luu^-T Luu^-1 = dpotrs(Luu, V^T)^T for the n rows of V (n x M row-major, in place; batched over Q with strides sV / sL):
// two BLOCKED TRIANGULAR SOLVES, 32-column diagonal blocks by true substitution (trsm_diag_kernel), the updates between them as
// GEMMs (alpha = -1, beta = 1) -- backward stable like LAPACK's dtrsm.  Used by the strict q(f) mode and hmogp_potrs_rows.
// `Vsrc` (optional, same layout as V): the right-hand sides; they reach V either by a copy in front of the solve or -- where every
// launch of the forward solve is one of the specialised ones -- through the FIRST touch of each column (no copy: 8.4 ms at H).
// [r6] `rdiag` (Q x M scratch) offers the solve to the one-launch-per-block kernels of trsm_panel.hip (needs `Lsym`, M a multiple of
// 128, n >= 1024); with `stats` the LAST direction's epilogue also leaves the row statistics sp = (result) . vec and
// sk = rowsum(result .* K) -- or rowsum(result .* result) when K is null -- (partials: launch_trsm_stats_combine).
// `dirs`: 1 = forward substitution only (V <- V Luu^-T), 2 = backward only (V <- V Luu^-1), 3 = both (dpotrs).
// `lsym_ready`: Lsym and rdiag already hold the images of THIS factor (the engine prepares them once per factorisation).
// Returns true iff the statistics were produced.
struct TrsmRowStats {
  const double* K = nullptr;     // same layout as V, or null
  const double* vec = nullptr;   // element (column j, batch q) at vec[q * vecB + j * vecS]
  long long vecB = 0, vecS = 1;
  double* part = nullptr;        // [Q][sPart]: [2 statistics][4 wave columns][ld]
  long long sPart = 0, ld = 0;
};
bool potrs_rows_inplace(double* V, long long sV, const double* Luu, long long sL, int M, long long n, int Q, hipStream_t st,
                        double* Lsym = nullptr, const double* Vsrc = nullptr, double* rdiag = nullptr,
                        const TrsmRowStats* stats = nullptr, int dirs = 3, bool lsym_ready = false);

// ------------------------------------------------------------------------------------ RCCL, resolved at run time
// The exchange step of a row-sharded run is ONE ncclAllReduce on the engine's own stream.  librccl is not a
// link-time dependency: a single-GPU user never needs it, and in a process that has already loaded a librccl.so.1 (PyTorch
// bundles one) dlopen() by soname returns THAT copy, so the library owns exactly one RCCL per process.
struct RcclApi {
  void* lib = nullptr;
  hm_nccl::GetUniqueId getUniqueId = nullptr;
  hm_nccl::CommInitRank commInitRank = nullptr;
  hm_nccl::CommDestroy commDestroy = nullptr;
  hm_nccl::CommAbort commAbort = nullptr;                   // optional (old builds): the watchdog degrades to an error return
  hm_nccl::CommGetAsyncError commGetAsyncError = nullptr;   // optional
  hm_nccl::AllReduce allReduce = nullptr;
  hm_nccl::GetErrorString getErrorString = nullptr;
  std::string why;
  bool ok() const { return lib != nullptr; }
};
RcclApi& rccl();
#define RCCL_TRY(expr)                                                                               \
  do {                                                                                               \
    int _r = (expr);                                                                                 \
    if (_r != hm_nccl::Success)                                                                           \
      throw EngineError{HMOGP_E_COMM, std::string("RCCL: ") + rccl().getErrorString(_r) + " in " #expr}; \
  } while (0)


}  // namespace hmogp_detail
using namespace hmogp_detail;


// =================================================================================================== engine
struct hmogp_engine {
  int T = 0, Q = 0, M = 0, P = 0, Df = 0, device = 0;
  long long chunk = 1048576;  // rows per pool (hmogp_config.chunk_rows); workspaces are sized by the rows actually streamed
  bool use_windows = false, cache_kuu = false, kuu_key_valid = false, no_small = false;
  // [r5] STRICT q(f) (HMOGP_CFG_STRICT_QF): q(f)'s mean and variance and the row-side gradient statistics are formed the way the
  // reference forms them -- A = K^ Kuu^-1 through two triangular factors of Luu (its dpotrs, svmogp_inf.py:214), v = ||L_q^T A^T||^2 -
  // A . K^ (:217-218), dVE_dmu = A^T alpha (:144), dVE_dS = A^T diag(beta) A (:145-148), dL_dKmn through A (S Kuu^-1 - I) (:157-161)
  // -- instead of through the explicit C_q = Kuu^-1 S Kuu^-1 - Kuu^-1, which differs from them by ~cond(Kuu) eps (1e-4 relative in
  // g_W / g_kappa / g_Z once GPy's jitter ladder is taken, cond ~ 1e7).  ~2x the step time at the headline size (2.5x in round 5, 3.3x in its first version); for parity in that regime.
  bool strict = false;         // ... of the CURRENT / last evaluation: strict_cfg (the config flag) or hmogp_params.eval_flags
  bool strict_cfg = false;
  DevBuf Dm, Ah, vpg, vcg, rdiag, trsmpart;
  // [r6] one-solve form: Wq = Luu^-1 L_q, Dm = Luu^-1 (S Kuu^-1 - I), w3 = Luu^-1 m (strict_stack / strict_unstack), Vst = the
  // stacked right-hand sides of that M x M solve and of the two that turn X^T diag(beta) X, X^T alpha into dVE_dS, dVE_dmu (finish)
  // Which form an evaluation takes: the ONE-solve form while the condition estimate of ITS OWN K_uu (variance max diag K_uu^-1, read
  // back behind the factorisation: one host wait the latency-bound chain covers) is <= 1e6 -- everything GPy's jitter rung 0 leaves
  // behind included --, the literal TWO-solve form of round 5 (A = dpotrs on the n x M side, D2 = S K_uu^-1 - I) beyond, where only
  // multiples of the reference's own sensitivity can be asserted and the one-solve P~ drifts out of them.  u_algebra (engine_rows.hip).
  bool strict_two = false, cond_two = false;   // cond_two: estimate beyond 1e6 (kept with a cached K_uu chain)
  double* h_cond = nullptr;    // page-locked landing buffer of the early condition estimate
  hipEvent_t ev_cond = nullptr;
  DevBuf D2, dcond;
  DevBuf Wq, w3, Vst, Lsy;
  long long sVst = 0;
  bool lsym_valid = false;     // Lsy / rdiag belong to the current Luu
  void strict_factor_images();
  unsigned quirks = HMOGP_QUIRKS_REFERENCE;
  std::vector<double> h_Z, kuu_key;
  std::vector<int> rung_request, kuu_rung;
  std::vector<int> f_index, d_index;
  std::vector<Task> tasks;
  hipStream_t st = nullptr;
  std::string err;

  // bundle layout (float64 words): [0] sum VE | [1] #(v<0) | [2,2+Df) sgv ; per q: H | r | dZ | sa | sl | swk
  long long NG = 0, per_q = 0, nstats = 0, oR = 0, oDZ = 0, oSA = 0, oSL = 0, oSWK = 0;

  // parameters of the current / last evaluation
  std::vector<double> h_var, h_ell, h_W, h_kap, h_W0, h_kap0, h_bs;
  std::vector<long long> rb, re;
  std::vector<int> rung;
  unsigned group_mask = HMOGP_GROUP_ALL;
  DevBuf dZ, dmu, dLflat, dvar, dell, dW, dkap, dsmall, dparams;
  double* h_small = nullptr;
  long long n_small = 0, oZ = 0, oMu = 0, oLf = 0, n_params = 0, oJit = 0, oW0 = 0, oBs = 0, oSeq = 0;
  int eval_seq = 0;            // evaluation counter of the small path (u_small_kernel's hand-over flags compare against it)
  // M x M (each Q*M*M)
  DevBuf Kuu, Luu, Kuui, L, S, KiS, KSK, C, Ctri, Sqi, tmpA, tmpB, HK, G, GSK, dKmm, dLdS;
  DevBuf a, Kr, gmu, gL, klout, rowout, dinfo, djit, dscr;
  // N x M workspaces and row vectors
  long long ws_rows = 0;
  DevBuf Kh, Pt, vp, vc, vpt, vct, valpha, vbeta, valpha0, vbeta0;
  DevBuf colred;               // [Q][M] per-column sums of E .* r2 (column statistics) before they are added into sl_q
  DevBuf stats, wire, slabs, colpart, quadpart, fwdpart, winrow, wincol, winhit, Xws, dstage;
  long long nwire = 0;  // float64 words of the wire format (lower triangles of H_q only; rowpass.hip: wire_tri_kernel)
  int* h_info = nullptr;     // page-locked landing buffer of the factorisation's info flags
  double* hstage = nullptr;  // page-locked landing buffer of the small per-evaluation results
  size_t hstage_cap = 0;
  bool began = false, evaluated = false;
  // device-resident q(u) for the SVI loop (hmogp_qu_*): dmu / dLflat ARE the parameters; Adadelta state beside them
  bool qu_resident = false;
  DevBuf ad_gms_m, ad_sms_m, ad_step_m, ad_pend_m, ad_gms_L, ad_sms_L, ad_step_L, ad_pend_L;

  // timing
  struct Span {
    hipEvent_t a, b;
    int cat;
  };
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  std::vector<Span> spans;
  double ms[NCAT] = {0};
  long long launches[NCAT] = {0};
  hipEvent_t ev_begin0 = nullptr, ev_begin1 = nullptr, ev_fin0 = nullptr, ev_fin1 = nullptr;
  bool st2_masked = false;    // the second stream leaves a few CUs of every XCD to the latency-bound chains (HMOGP_ST2_FREE)
  // [r4] SMALL-PROBLEM MODE (M <= 128 and <= 65536 rows in the evaluation; BASELINE config 1 is M = 50, 3000 rows): such a step is
  // bound by the HOST (45 launches, 15 copies, 37 event records: ~0.6 ms of API time, profiles/r04_C1_hip_api_stats_before.csv) and by
  // cross-queue dependencies (every hipStreamWaitEvent between two hardware queues costs ~10 us of device idle time), not by
  // any kernel.  In this mode the three streams are ONE (st2 = st3 = st: event waits on the same queue are free) and the per-
  // family timing spans are not recorded (hmogp_last_timings then reports the total only).
  hipStream_t st2_own = nullptr, st3_own = nullptr;
  bool small_mode = false;
  // ... and with M <= HMOGP_SMALL_M the replicated M x M algebra runs as TWO fused kernels, one block per latent with every matrix
  // in LDS (small_model.hip), instead of ~30 launches.  A factorisation that needs GPy's jitter ladder is repeated on the regular
  // path (small_veto), which owns the ladder.
  bool small_path = false, small_veto = false, small_info_pending = false, info_early = false;
  double* hstage_dev = nullptr;   // hstage as the device addresses it (finish_small_kernel writes the results there itself)
  bool small_rows = false;     // ... and its row pass as the two fused kernels of small_model.hip (small_fwd / small_bwd)
  DevBuf smallslab;
  struct RetryRegular {};
  // [r4] hipGraph of one small-model evaluation.  The small path is a FIXED sequence on one stream (one upload from the page-locked
  // parameter image, ~20 kernels, one download into the page-locked staging block) whose kernel arguments do not depend on the
  // parameter VALUES (quad_kernel reads the mixing weights from the parameter block): the second evaluation with the same key
  // (gradient gates, row ranges, forced rungs, resident q(u) or not) is captured, every later one is a replay -- one
  // hipGraphLaunch instead of ~35 API calls.  Graphs are dropped when the data or the workspaces change.
  struct SmallGraph {
    std::vector<long long> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
  };
  std::vector<SmallGraph> graphs;
  std::vector<std::vector<long long>> warm_keys;
  long long graph_replays = 0, graph_captures = 0;
  void drop_graphs(bool keep_warm = false);
  std::vector<long long> graph_key(const hmogp_params* p) const;
  // hmogp_elbo_grad on the small path: normal evaluation the first time a key is seen, capture + launch the second time, replay
  // afterwards.  Returns false when the call does not qualify (the caller then runs the normal begin / finish).
  bool graph_step(const hmogp_params* p, hmogp_outputs* out);
  bool graphs_broken = false, via_graph = false;
  std::vector<long long> pending_warm;
  void mark_warm();
  hipStream_t st2 = nullptr;  // second stream, LOW priority: bandwidth-bound work beside the main stream (K_uf prefetch, colstats)
  hipStream_t st3 = nullptr;  // third stream, HIGH priority like the main one: the q(u)-only chains (S, S^-1; dL/dL, D2H)
  hipEvent_t ev_qu = nullptr;   // behind an in-place update of the resident q(u) (hmogp_qu_natgrad)
  hipEvent_t ev_fork = nullptr, ev_gsk = nullptr, ev_zero = nullptr, ev_info = nullptr, ev_S = nullptr, ev_join = nullptr, ev_col = nullptr, ev_kuf = nullptr, ev_params = nullptr,
             ev_ua = nullptr;

  hipEvent_t new_event();
  struct Scope {
    hmogp_engine* e;
    Span s;
    hipStream_t stream;
    bool on_;
    Scope(hmogp_engine* eng, int cat, int nlaunch, hipStream_t on = nullptr) : e(eng), stream(on ? on : eng->st) {
      static const bool off = getenv("HMOGP_NO_SPANS") != nullptr;   // experiment: what the timing events themselves cost
      on_ = !off && (!eng->small_mode || cat == CAT_EXCHANGE);   // (the exchange step is always timed)
      s.cat = cat;
      e->launches[cat] += nlaunch;
      if (!on_) return;
      s.a = e->new_event();
      s.b = e->new_event();
      (void)hipEventRecord(s.a, stream);
    }
    ~Scope() {
      if (!on_) return;
      (void)hipEventRecord(s.b, stream);
      e->spans.push_back(s);
    }
  };
  void collect_spans();

  // ---- native exchange step (hmogp_comm_*): one RCCL communicator per engine, collectives on the engine's stream ----
  ncclComm_t comm = nullptr;
  int comm_ranks = 1, comm_rank = 0;
  bool exchanged = false;      // the bundle of the current step has been all-reduced

  void comm_init(int nranks, int rank, const void* id);
  void comm_destroy();
  // A rank that cannot contribute to the step's collective (its row pass failed: HIP OOM, bad row range, E_STATE ...) ABORTS
  // the communicator, so that the peers' ncclAllReduce ends with an error instead of blocking for ever; the engine
  // is left without a communicator (hmogp_comm_info: 0 ranks) and every later sharded call fails with HMOGP_E_STATE.
  void comm_abort();
  // Wait for the engine's stream while a collective is in flight: the torch path this replaces has a watchdog, RCCL alone has
  // none.  Polls the stream, the communicator's asynchronous error state and a deadline (HMOGP_COMM_TIMEOUT_S, default 600 s;
  // 0 = wait for ever); on either failure the communicator is aborted and HMOGP_E_COMM is reported.
  void wait_exchanged();
  // pack -> ncclAllReduce(sum, fp64, in place on the wire buffer) -> unpack, all ENQUEUED on the engine's stream: no host
  // synchronisation, no other library's stream.  The wire format holds the lower triangles of H_q only (12.7 MB instead
  // of 25.2 MB at M = 1024, Q = 3).
  void exchange();

  ~hmogp_engine();

  double* Hq(int q) { return stats.d() + NG + q * per_q; }

  void init(const hmogp_config* c);

  void set_task_data(int t, const double* X, const double* Y, long long N);

  void ensure_workspace(long long rows);

  // the strict mode's own buffers, allocated when an evaluation first runs in that mode (config flag or per-evaluation flag)
  long long ws_strict_rows = 0;
  void ensure_strict_workspace();

  // ------------------------------------------------------------------------------------------ parameters
  void upload_params(const hmogp_params* p, bool enqueue = true);

  // batched (over q) M x M GEMM helper
  void mm(const double* A, bool a_k, const double* B, bool b_k, double* Cc, double alpha = 1.0, long long sA = -1,
          int lda = -1, hipStream_t stream = nullptr, int a_tri = 0, int b_tri = 0, bool lower_only = false) {
    GemmArgs g;
    const long long MM = (long long)M * M;
    g.A = A, g.B = B, g.C = Cc;
    g.M = g.N = g.K = M;
    g.lda = lda > 0 ? lda : M, g.ldb = g.ldc = M;
    g.nbatch = Q;
    g.sA = sA >= 0 ? sA : MM, g.sB = g.sC = MM;
    g.a_kmajor = a_k, g.b_kmajor = b_k;
    g.alpha = alpha;
    g.a_tri = a_tri, g.b_tri = b_tri;
    g.lower_only = lower_only ? 1 : 0;
    launch_gemm_f64(g, stream ? stream : st);
  }

  // ------------------------------------------------------------------------------------------ u algebra
  void u_algebra_small();

  void u_algebra();

  // ------------------------------------------------------------------------------------------ row pools
  // Rows are streamed in POOLS of at most `chunk` rows.  Everything between the covariance construction and the
  // quadrature, and everything after it, is independent of which task a row belongs to (K^ C_q, the row statistics,
  // the weighted Gram and the column statistics only see rows), so a pool concatenates row ranges ("segments") of
  // consecutive tasks: one forward contraction, one Gram product and one column-statistics pass per pool instead of
  // one per task -- fewer, larger launches (tails, launch-bound reductions; matters most for minibatches and for
  // the per-rank shares of a multi-GPU run).  Only K_uf construction and the quadrature run per segment.  The
  // exact-zero windows need spatially sorted rows per launch, so that mode keeps one task per pool.
  struct Seg { int t; long long r0, n, off; };
  std::vector<std::vector<Seg>> pools;
  bool kuf_prefetched = false;
  // K_uf is built in launches of KUF_CHUNK_ROWS rows on the side stream (see kuf_pool); the forward contraction is ONE launch
  // per pool.  (Measured alternative: one forward launch per task segment, each waiting only for its own part of K_uf --
  // 127.3 vs 126.3 ms at the headline size, 34.15 vs 33.7 ms at 50 000 rows per task: K_uf construction beside a forward
  // costs the forward what it takes alone, and every extra launch adds a partially filled last round of blocks.)
  static constexpr long long KUF_CHUNK_ROWS = 16384;
  void plan_pools();
  // K_uf = k_q(X, Z_q) of one pool, all latents in one launch per segment (grid.z = latent), on `stream`
  void kuf_pool(const std::vector<Seg>& pl, hipStream_t stream, size_t seg_begin = 0, size_t seg_end = (size_t)-1);

  // inputs of a multi-segment pool, contiguous in pool order (fs_x of the forward epilogue, the column statistics)
  // (the staged copy is reused while the SAME segments of the SAME data are asked for again -- every full-batch evaluation after
  //  the first: one D2D copy per task less per step, which matters for host-bound small models)
  std::vector<long long> staged_key;
  void stage_pool_inputs(const std::vector<Seg>& pl, hipStream_t stream);

  // strict q(f): the solve-based forms of svmogp_inf.py:212-218 for the n pool rows whose K^ sits in Kh (all latents batched).
  // A = K^ Kuu^-1 = dpotrs(Luu, K^T)^T by two BLOCKED TRIANGULAR SOLVES against Luu: 32-column diagonal blocks by true substitution
  // (trsm_diag_kernel), the updates between them as GEMMs -- backward stable like LAPACK's dtrsm.  (Round 5 first used two
  // products with the explicit Luu^-1: m_fd was then 9e-8 of its scale away from the reference at cond(K_uu) = 1e7, 1.4e-2 at
  // cond 1e12; with the substitution 3e-10 / the reference's own rounding sensitivity.)
  void strict_forward(long long n, const double* X, bool grads, bool hyper);

  // ------------------------------------------------------------------------------------------ row pass
  void row_pass();

  // bundle <-> wire format, synchronous at return (the caller's all-reduce runs on another stream / library)
  void wire_copy(int dir);

  void decide_mode(const hmogp_params* p);

  bool sharded_call = false;
  bool skip_g_L = false;        // HMOGP_EVAL_NO_G_L of this evaluation
  void begin(const hmogp_params* p, bool sync = true, bool will_exchange = false);
  // after a synchronisation behind u_small_kernel: did a latent's plain factorisation fail?  (forced rung: an error)
  bool small_failed();

  // ------------------------------------------------------------------------------------------ finish
  // what one evaluation returns through the single D2H staging block: the small results (head of the bundle, KL partials,
  // per-latent tails, K_uu-side rows) gathered device-side; on the small-model path the q(u) gradients ride in the same block
  struct FinLayout {
    bool want_qu = false, want_hz = false, qu_out = false;
    size_t n_hg = 0, n_kl = 0, n_tail = 0, n_row = 0, n_all = 0, n_gmu = 0, n_gl = 0, n_stage = 0;
  } fl;
  void fin_layout(const hmogp_outputs* out);
  void finish(hmogp_outputs* out);
  void finish_enqueue(hmogp_outputs* out);
  // everything behind the last enqueued operation of an evaluation: the one host synchronisation, then the host assembly
  void finish_tail(hmogp_outputs* out);

  // ------------------------------------------------------------------------------------------ consumers
  void posterior_u(double* wv, double* winv);

  // ---- device-resident q(u) + Adadelta (SVI loop; util.py:321-329, svmogp.py:188-199) ------------
  void qu_load(const double* m_u, const double* L_flat);
  void qu_read(double* m_u, double* L_flat);
  // phase 0: momentum move before the gradient evaluation; phase 1: update from the gradients the last evaluation left in
  // gmu / gL (objective = -ELBO: sign -1), or from a zero gradient when that evaluation did not include the q(u) group
  void qu_adadelta(int phase, double rate, double m, double d, double omd, double o);

  // Inner-protocol debug export (include/hetmogp_hip.h: hmogp_debug_raw_grads): the gradient dictionary of
  // SVMOGPInf.inference (svmogp_inf.py:107) rebuilt from what the last evaluation left in HBM -- dKmm, a, P~ of the one
  // pool, p / c row statistics -- plus one more quadrature pass that writes the per-function d ve/dm, d ve/dv rows.
  void debug_raw(double* o_kmm, double* o_kmn, double* o_kdiag);

  // Natural-gradient update of q(u_q) = N(m_q, S_q) from the gradients of the last evaluation (the
  // north-star names it, the reference has none):  S^-1 <- S^-1 - 2 gamma dL/dS ;  S^-1 m <- S^-1 m + gamma (dL/dm -
  // 2 dL/dS m) ;  then m and L = chol(S) are recovered.  Requires the q(u) group in the last evaluation's mask.
  bool have_qu_grads = false;
  DevBuf ng_t1, ng_t2, ng_th, ng_mnew, ng_mq, ng_lflat;
  int* h_info2 = nullptr;   // page-locked: the two factorisations' info words of a natural-gradient step
  // Core of the natural-gradient step: leaves the new m_u ([M, Q], the layout of dmu) in ng_mq and the new packed Cholesky
  // factor in ng_lflat; throws HMOGP_E_NOT_PD (nothing modified) when the step leaves the positive-definite cone.  ONE host
  // synchronisation (the two info words) at the end; everything else is enqueued back to back on the engine's stream.
  void natgrad_core(double gamma, bool sync = true);
  // Natural-gradient update of q(u_q) = N(m_q, S_q) from the gradients of the last evaluation (the
  // north-star names it, the reference has none):  S^-1 <- S^-1 - 2 gamma dL/dS ;  S^-1 m <- S^-1 m + gamma (dL/dm -
  // 2 dL/dS m) ;  then m and L = chol(S) are recovered.  Requires the q(u) group in the last evaluation's mask.
  void natgrad_step(double gamma, double* m_out, double* L_flat_out);
  // The same step on the DEVICE-RESIDENT q(u) (hmogp_qu_load): m_u / L_flat are updated in place in HBM, nothing but the two
  // info words crosses PCIe -- the natural-gradient SVI loop (E-steps) without moving 2 x 12.6 MB per iteration.
  void qu_natgrad(double gamma);

  // [r5, ABI v7] hmogp_qu_natgrad without the host synchronisation: the commit into the resident q(u) is conditional ON THE DEVICE
  // (commit_if_ok_kernel reads the factorisation's info words), the call returns with everything enqueued, and the caller goes
  // straight on to the next evaluation -- whose parameter upload, pool staging and K_uf construction (second stream) then run BESIDE
  // this step's latency-bound factorisation chain instead of behind a host round trip.  hmogp_qu_natgrad_status waits and reports.
  hipEvent_t ev_ng = nullptr;
  bool ng_pending = false, ng_last_taken = true;
  void qu_natgrad_async(double gamma);
  int qu_natgrad_status();

  void predict_f(const double* Xnew, long long Nnew, double* m, double* v);
};
