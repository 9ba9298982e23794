// engine.hip -- host side of the hetmogp HIP engine and its C ABI (include/hetmogp_hip.h).
//
// One evaluation = SVMOGP.parameters_changed() (hetmogp/svmogp.py:85-166):
//   u_algebra   replicated M x M work before the rows are touched: K_uu, jitchol ladder, K_uu^-1, S = L L^T,
//               a = K_uu^-1 m, C = K_uu^-1 S K_uu^-1 - K_uu^-1, S^-1            (util.py:181-200, svmogp_inf.py:192-195)
//   row_pass    per task, per row chunk:  K^ = k_q(X, Z_q)  ->  P~ = K^ C_q (FP64 MFMA)  ->  p, c row statistics
//               ->  q(f), variational expectations, row weights  ->  H_q += K^T diag(beta) K^ (FP64 MFMA),
//               r_q, dZ column statistics.  Everything lands in ONE additive statistic bundle (the only thing
//               a row-sharded multi-GPU run has to all-reduce).
//   finish      replicated M x M post-processing of the bundle: svmogp_inf.py:111-183,227-250 and the
//               parameter-gradient assembly of svmogp.py:101-166.
// There is no CPU fallback anywhere in this file.
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

namespace {

struct EngineError {
  int code;
  std::string msg;
};

thread_local std::string g_create_error;

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

enum { CAT_TOTAL = 0, CAT_RBF, CAT_FWD, CAT_ROWSTATS, CAT_QUAD, CAT_GRAM, CAT_COLSTATS, CAT_MM, CAT_EXCHANGE, NCAT };
static_assert(NCAT == HMOGP_NTIMINGS, "hmogp_last_timings layout");

struct Task {
  long long N = 0;
  DevBuf X, Y, Yaux;
  int lik = 0, dimf = 1, d0 = 0;
  double param = 0.0;
  DevBuf offsets;  // device: quad scalar slot -> bundle offset
  int nscal = 0;
};

int lik_dimf(int lik, double param) {
  switch (lik) {
    case HMOGP_LIK_GAUSSIAN:
    case HMOGP_LIK_BERNOULLI:
    case HMOGP_LIK_POISSON:
    case HMOGP_LIK_EXPONENTIAL: return 1;
    case HMOGP_LIK_HETGAUSSIAN:
    case HMOGP_LIK_GAMMA:
    case HMOGP_LIK_BETA: return 2;
    case HMOGP_LIK_CATEGORICAL: return (int)param - 1;
    default: return -1;
  }
}

// Row ranges per weighted-Gram launch: a multiple of 8 (one range per XCD at a time), each >= 32 k-steps of 16 rows,
// enough blocks (lower tiles x ranges) for >= 8 rounds over the 256 CUs, at most KS_MAX slabs.
constexpr int KS_MAX = 256;  // most row ranges (slabs) of the weighted Gram
// rows per slab of the column statistics: 256, or 32 for short passes (one thread owns two columns and walks the rows of its
// slab one after the other: at M = 50 a 256-row slab is 25 threads x 256 dependent steps, 110 us for 3000 rows)
inline long long col_split(long long n) { return n <= 16384 ? 32 : 256; }
int gram_ksplit(long long n, int M) {
  static const int forced = [] {   // HMOGP_GRAM_KSPLIT=<row ranges> (experiments; profiles/r03_gram_ksplit.txt: flat)
    const char* e = getenv("HMOGP_GRAM_KSPLIT");
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) return (int)std::max<long long>(1, std::min<long long>(forced, (n + 15) / 16));
  const int tiles = (M + 127) / 128, ntl = tiles * (tiles + 1) / 2;
  const long long ksteps = (n + 15) / 16;
  // enough blocks to fill the chip several times over, and row ranges of at most ~8192 rows (the tiles of one range drift
  // apart as they stream it; shorter ranges keep the shared K^ rows in that XCD's L2)
  // [r5] ... but every range costs a slab that reduce_slabs_lower has to stream again (M = 1024, Q = 3: 14 MB per range, 147 us for
  // 56 of them behind the Gram of an 8192-row minibatch step): short passes take ranges of >= 2048 rows as long as >= 4.5 rounds
  // of blocks remain (M = 1024: 32 ranges instead of 56 at 4 x 8192 rows: 7.67 -> 7.59 ms per step; M >= 2048 and the
  // full-batch sizes are unchanged)
  const long long floor8 = (((4 * 256 + 128 + ntl - 1) / ntl + 7) / 8) * 8;
  long long want = std::max<long long>(std::max<long long>(floor8, std::min<long long>((8 * 256 + ntl - 1) / ntl, n / 2048)), n / 8192);
  want = std::min<long long>(std::min<long long>(KS_MAX, std::max<long long>(1, ksteps / 32)), want);
  // [r4] short passes (a few thousand rows, small M: BASELINE config 1): a handful of blocks each looping over hundreds of rows
  // is latency-bound (77 us for 3000 rows at M = 50) -- row ranges of 8 k-steps while the grid stays below one block per CU
  if (ntl * want < 256) want = std::max(want, std::min<long long>(std::min<long long>(KS_MAX, std::max<long long>(1, ksteps / 8)), 256 / ntl));
  return (int)(want >= 8 ? (want / 8) * 8 : std::max<long long>(1, want));
}

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
                     double* d_jit, double* dscr, hipStream_t st, JitcholState& js, int part = -1) {
  if (part == 1) {
    launch_potrf_batched(Luu, Q, M, d_info, dscr, st, JIT_HEAD_PANELS, -1);
    HIP_TRY(hipMemcpyAsync(js.info, d_info, sizeof(int) * Q, hipMemcpyDeviceToHost, st));
    return;
  }
  js.jit.assign(Q, 0.0), js.forced.resize(Q);
  if (!js.info) js.info_own.assign(Q, 0), js.info = js.info_own.data();
  for (int q = 0; q < Q; ++q) {
    js.forced[q] = rung_io[q] != -2;
    if (rung_io[q] >= 0) js.jit[q] = diag_mean[q] * 1e-6 * std::pow(10.0, rung_io[q]);
    if (!js.forced[q]) rung_io[q] = -1;
  }
  HIP_TRY(hipMemcpyAsync(d_jit, js.jit.data(), sizeof(double) * Q, hipMemcpyHostToDevice, st));
  launch_add_diag_copy(Kuu, Luu, Q, M, d_jit, st);
  if (part == 0 && (M + HMOGP_POTRF_NB - 1) / HMOGP_POTRF_NB > JIT_HEAD_PANELS) {
    launch_potrf_batched(Luu, Q, M, d_info, dscr, st, 0, JIT_HEAD_PANELS);
    return;
  }
  launch_potrf_batched(Luu, Q, M, d_info, dscr, st);
  HIP_TRY(hipMemcpyAsync(js.info, d_info, sizeof(int) * Q, hipMemcpyDeviceToHost, st));
  js.complete = true;
}
void jitchol_resolve(const double* Kuu, double* Luu, int Q, int M, const double* diag_mean, int* rung_io, int* d_info,
                     double* d_jit, double* dscr, hipStream_t st, JitcholState& js) {
  HIP_TRY(hipStreamSynchronize(st));
  const long long MM = (long long)M * M;
  for (int q = 0; q < Q; ++q) {
    if (js.info[q] == 0) continue;
    if (js.forced[q]) throw EngineError{HMOGP_E_NOT_PD, "Cholesky failed at the forced jitter rung"};
    if (!(diag_mean[q] > 0.0)) throw EngineError{HMOGP_E_NOT_PD, "not pd: non-positive diagonal elements"};
    double j = diag_mean[q] * 1e-6;
    bool ok = false;
    for (int k = 0; k < 5 && std::isfinite(j); ++k, j *= 10.0) {
      HIP_TRY(hipMemcpyAsync(d_jit, &j, sizeof(double), hipMemcpyHostToDevice, st));
      launch_add_diag_copy(Kuu + q * MM, Luu + q * MM, 1, M, d_jit, st);
      launch_potrf_batched(Luu + q * MM, 1, M, d_info, dscr, st);
      int inf1 = 0;
      HIP_TRY(hipMemcpyAsync(&inf1, d_info, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (inf1 == 0) {
        rung_io[q] = k;
        ok = true;
        break;
      }
    }
    if (!ok) throw EngineError{HMOGP_E_NOT_PD, "not positive definite, even with jitter."};
  }
}
void jitchol_batched(const double* Kuu, double* Luu, int Q, int M, const double* diag_mean, int* rung_io, int* d_info,
                     double* d_jit, double* dscr, hipStream_t st) {
  JitcholState js;
  jitchol_enqueue(Kuu, Luu, Q, M, diag_mean, rung_io, d_info, d_jit, dscr, st, js);
  jitchol_resolve(Kuu, Luu, Q, M, diag_mean, rung_io, d_info, d_jit, dscr, st, js);
}

// V <- V Luu^-T Luu^-1 = dpotrs(Luu, V^T)^T for the n rows of V (n x M row-major, in place; batched over Q with strides sV / sL):
// two BLOCKED TRIANGULAR SOLVES, 32-column diagonal blocks by true substitution (trsm_diag_kernel), the updates between them as
// GEMMs (alpha = -1, beta = 1) -- backward stable like LAPACK's dtrsm.  Used by the strict q(f) mode and hmogp_potrs_rows.
// `Vsrc` (optional, same layout as V): the right-hand sides; they reach V either by a copy in front of the solve or -- where every
// launch of the forward solve is one of the specialised ones -- through the FIRST touch of each column (no copy: 8.4 ms at H).
void potrs_rows_inplace(double* V, long long sV, const double* Luu, long long sL, int M, long long n, int Q, hipStream_t st,
                        double* Lsym = nullptr, const double* Vsrc = nullptr) {
  // C[:, c0:c0+nc] -= V[:, a0:a0+k] op(B)   (op(B) = Luu[c0.., a0..]^T for the forward solve, Luu[a0.., c0..] for the backward one)
  // [r5] `role` 1 offers the update to the specialised 8-wave kernel (gemm_rowpass.hip, C -= A B form: 128-column updates with a
  // k-major B); it falls back to the general kernel by itself.  The forward solve's B is the TRANSPOSE of a block of Luu: with
  // `Lsym` (a Q x M x M scratch) it is read k-major from a mirrored copy of the factor.
  auto update_args = [&](int c0, int nc, int a0, int k, const double* B_, int b_kmajor) {
    GemmArgs g;
    g.A = V + a0, g.lda = M, g.a_kmajor = 0, g.sA = sV;
    g.B = B_, g.ldb = M, g.b_kmajor = b_kmajor, g.sB = sL;
    g.C = V + c0, g.ldc = M, g.sC = sV;
    g.M = (int)n, g.N = nc, g.K = k;
    g.alpha = -1.0, g.beta = 1.0;
    g.nbatch = Q;
    g.role = 1;
    return g;
  };
  auto update = [&](int c0, int nc, int a0, int k, const double* B_, int b_kmajor, const double* c_src = nullptr) {
    GemmArgs g = update_args(c0, nc, a0, k, B_, b_kmajor);
    g.c_src = c_src ? c_src + c0 : nullptr;
    launch_gemm_rowpass_or_general(g, st);
  };
  const bool sym = Lsym != nullptr && n >= 4096 && sL == (long long)M * M;
  if (sym) {
    HIP_TRY(hipMemcpyAsync(Lsym, Luu, sizeof(double) * sL * Q, hipMemcpyDeviceToDevice, st));
    launch_mirror_lower(Lsym, Q, M, sL, st);       // Lsym[k][j] = Luu[j][k] above the diagonal
  }
  // Two-level blocking: the bulk of the flops sits in updates of 128 columns at a time (full MFMA tiles: an update of a 32-column
  // block alone uses a quarter of a 128 x 128 tile), the 32-column substitution steps and their short updates stay inside a
  // 128-column block (strict forward at the headline size: 568 ms one-level, 337 ms two-level).
  // [r5] inside a 128-column block the short updates ride in the substitution launches (right-looking, on the matrix cores, x taken
  // from LDS: trsm_diag_kernel) where the shape allows it: 7 launches and ~740 column passes over HBM per block become 4 and 640
  static const bool fuse_env = [] {   // HMOGP_TRSM_FUSE=0: separate 32-column GEMM updates (A/B runs)
    const char* e = getenv("HMOGP_TRSM_FUSE");
    return !(e && e[0] == '0');
  }();
  const bool fuse = fuse_env && trsm_diag_can_fuse(V, sV, M);
  constexpr int NB = 128;
  // first touch instead of a copy: every 128-column update of the forward solve must be taken by the specialised kernel (the
  // general one has no separate source) and every block's first substitution launch by the row-coalesced one
  bool first_touch = false;
  if (Vsrc) {
    static const bool ft_env = [] {   // HMOGP_STRICT_FIRST_TOUCH=0: copy the right-hand sides in front of the solve (A/B runs)
      const char* e = getenv("HMOGP_STRICT_FIRST_TOUCH");
      return !(e && e[0] == '0');
    }();
    first_touch = ft_env && fuse && sym && (M % NB) == 0 && (reinterpret_cast<uintptr_t>(Vsrc) & 15) == 0;
    for (int J0 = NB; first_touch && J0 < M; J0 += NB) {
      GemmArgs g = update_args(J0, NB, 0, J0, Lsym + J0, 1);
      g.c_src = Vsrc + J0;
      first_touch = gemm_rowpass_would_take(g);
    }
    if (!first_touch)
      for (int q = 0; q < Q; ++q)
        HIP_TRY(hipMemcpyAsync(V + q * sV, Vsrc + q * sV, sizeof(double) * n * M, hipMemcpyDeviceToDevice, st));
  }
  for (int J0 = 0; J0 < M; J0 += NB) {                        // X Luu^T = V   (forward over the columns)
    const int J1 = std::min(M, J0 + NB);
    if (J0 > 0) {
      if (sym) update(J0, J1 - J0, 0, J0, Lsym + J0, 1, first_touch ? Vsrc : nullptr);   // op(B)[k][j] = Luu[J0 + j][k] = Lsym[k][J0 + j]
      else update(J0, J1 - J0, 0, J0, Luu + (long long)J0 * M, 0);
    }
    for (int j0 = J0; j0 < J1; j0 += 32) {
      const int nb = std::min(32, J1 - j0);
      if (fuse) {
        launch_trsm_diag(0, V, sV, Luu, sL, M, j0, nb, n, Q, st, j0 + 32, J1, (first_touch && j0 == 0) ? Vsrc : nullptr);
        continue;
      }
      if (j0 > J0) update(j0, nb, J0, j0 - J0, Luu + (long long)j0 * M + J0, 0);
      launch_trsm_diag(0, V, sV, Luu, sL, M, j0, nb, n, Q, st);
    }
  }
  for (int J0 = ((M - 1) / NB) * NB; J0 >= 0; J0 -= NB) {     // A Luu = X     (backward over the columns)
    const int J1 = std::min(M, J0 + NB);
    if (J1 < M) update(J0, J1 - J0, J1, M - J1, Luu + (long long)J1 * M + J0, 1);
    for (int j0 = J0 + ((J1 - J0 - 1) / 32) * 32; j0 >= J0; j0 -= 32) {
      const int nb = std::min(32, J1 - j0), j1 = j0 + nb;
      if (fuse) {
        launch_trsm_diag(1, V, sV, Luu, sL, M, j0, nb, n, Q, st, J0, j0);
        continue;
      }
      if (j1 < J1) update(j0, nb, j1, J1 - j1, Luu + (long long)j1 * M + j0, 1);
      launch_trsm_diag(1, V, sV, Luu, sL, M, j0, nb, n, Q, st);
    }
  }
}

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
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
      const char* e = dlerror();
      a.why = e ? e : "dlopen failed";
    }
    if (!a.lib) return a;
    a.getUniqueId = (decltype(a.getUniqueId))dlsym(a.lib, "ncclGetUniqueId");
    a.commInitRank = (decltype(a.commInitRank))dlsym(a.lib, "ncclCommInitRank");
    a.commDestroy = (decltype(a.commDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.allReduce = (decltype(a.allReduce))dlsym(a.lib, "ncclAllReduce");
    a.getErrorString = (decltype(a.getErrorString))dlsym(a.lib, "ncclGetErrorString");
    a.commAbort = (decltype(a.commAbort))dlsym(a.lib, "ncclCommAbort");
    a.commGetAsyncError = (decltype(a.commGetAsyncError))dlsym(a.lib, "ncclCommGetAsyncError");
    if (!a.getUniqueId || !a.commInitRank || !a.commDestroy || !a.allReduce || !a.getErrorString) {
      a.why = "librccl is missing one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
      a.lib = nullptr;
    }
    return a;
  }();
  return api;
}
#define RCCL_TRY(expr)                                                                               \
  do {                                                                                               \
    int _r = (expr);                                                                                 \
    if (_r != hm_nccl::Success)                                                                           \
      throw EngineError{HMOGP_E_COMM, std::string("RCCL: ") + rccl().getErrorString(_r) + " in " #expr}; \
  } while (0)

}  // namespace

// =================================================================================================== engine
struct hmogp_engine {
  int T = 0, Q = 0, M = 0, P = 0, Df = 0, device = 0;
  long long chunk = 1048576;  // rows per pool (hmogp_config.chunk_rows); workspaces are sized by the rows actually streamed
  bool use_windows = false, cache_kuu = false, kuu_key_valid = false, no_small = false;
  // [r5] STRICT q(f) (HMOGP_CFG_STRICT_QF): q(f)'s mean and variance and the row-side gradient statistics are formed the way the
  // reference forms them -- A = K^ Kuu^-1 through two triangular factors of Luu (its dpotrs, svmogp_inf.py:214), v = ||L_q^T A^T||^2 -
  // A . K^ (:217-218), dVE_dmu = A^T alpha (:144), dVE_dS = A^T diag(beta) A (:145-148), dL_dKmn through A (S Kuu^-1 - I) (:157-161)
  // -- instead of through the explicit C_q = Kuu^-1 S Kuu^-1 - Kuu^-1, which differs from them by ~cond(Kuu) eps (1e-4 relative in
  // g_W / g_kappa / g_Z once GPy's jitter ladder is taken, cond ~ 1e7).  ~3.3x the step time at the headline size; for parity in that regime.
  bool strict = false;         // ... of the CURRENT / last evaluation: strict_cfg (the config flag) or hmogp_params.eval_flags
  bool strict_cfg = false;
  DevBuf Dm, Ah, vpg, vcg;
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
  void drop_graphs(bool keep_warm = false) {
    for (auto& g : graphs) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    graphs.clear();
    if (!keep_warm) warm_keys.clear();
  }
  std::vector<long long> graph_key(const hmogp_params* p) const {
    std::vector<long long> k{(long long)p->group_mask, (!p->m_u && !p->L_flat) ? 1 : 0};
    for (int t = 0; t < T; ++t) k.push_back(p->row_begin ? p->row_begin[t] : 0), k.push_back(p->row_end ? p->row_end[t] : tasks[t].N);
    for (int q = 0; q < Q; ++q) k.push_back(p->forced_rung ? p->forced_rung[q] : -2);
    return k;
  }
  // hmogp_elbo_grad on the small path: normal evaluation the first time a key is seen, capture + launch the second time, replay
  // afterwards.  Returns false when the call does not qualify (the caller then runs the normal begin / finish).
  bool graph_step(const hmogp_params* p, hmogp_outputs* out) {
    static const bool enabled = [] {   // HMOGP_SMALL_GRAPH=0: no graphs (A/B runs)
      const char* e = getenv("HMOGP_SMALL_GRAPH");
      return !(e && e[0] == '0');
    }();
    if (!enabled || !p || !out || out->dL_dS || small_veto || comm) return false;
    HIP_TRY(hipSetDevice(device));
    decide_mode(p);
    if (!small_path) return false;
    const std::vector<long long> key = graph_key(p);
    SmallGraph* hit = nullptr;
    for (auto& g : graphs)
      if (g.key == key) hit = &g;
    if (!hit) {
      if (std::find(warm_keys.begin(), warm_keys.end(), key) == warm_keys.end()) {
        pending_warm = key;          // first sight: a normal evaluation sizes every workspace; warm once it has SUCCEEDED
        return false;
      }
      if (graphs.size() >= 32) drop_graphs();
      // ---- capture: the normal code path, recorded instead of executed ------------------------------------------------------
      SmallGraph g;
      g.key = key;
      HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      bool ok = true;
      std::string why;
      try {
        begin(p, false);
        if (!out) throw EngineError{HMOGP_E_INVALID, "null outputs"};
        finish_enqueue(out);
      } catch (const EngineError& e) {
        ok = false, why = e.msg;
      } catch (const HipError& e) {
        ok = false, why = hipGetErrorString(e.code);
      }
      const hipError_t ec = hipStreamEndCapture(st, &g.graph);
      if (!ok || ec != hipSuccess || !g.graph) {
        if (g.graph) (void)hipGraphDestroy(g.graph);
        (void)hipGetLastError();
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
          throw EngineError{HMOGP_E_NO_DEVICE, "a failed hipGraph capture left the engine's stream in capture mode: " + why};
        (void)hipGetLastError();
        began = false;
        warm_keys.clear();            // (do not try again for this engine's current keys; the normal path reports real errors)
        graphs_broken = true;
        return false;
      }
      if (hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(g.graph);
        (void)hipGetLastError();
        began = false;
        graphs_broken = true;
        return false;
      }
      graphs.push_back(g);
      hit = &graphs.back();
      ++graph_captures;
    } else {
      // ---- replay: only the HOST side of begin() (validation, parameter image, pool plan, output layout) ----------------------
      began = false, exchanged = false;
      spans.clear(), pool_used = 0;
      for (int c = 0; c < NCAT; ++c) ms[c] = 0.0, launches[c] = 0;
      upload_params(p, false);
      plan_pools();
      kuu_key_valid = false;
      for (int q = 0; q < Q; ++q)
        if (rung[q] == -2) rung[q] = -1;
      small_info_pending = true;
      info_early = false;            // (the captured evaluation delivers its info words with the results)
      began = true;
      fin_layout(out);
      ++graph_replays;
    }
    HIP_TRY(hipGraphLaunch(hit->exec, st));
    via_graph = true;
    try {
      finish_tail(out);
    } catch (...) {
      via_graph = false;
      throw;
    }
    via_graph = false;
    return true;
  }
  bool graphs_broken = false, via_graph = false;
  std::vector<long long> pending_warm;
  void mark_warm() {
    if (!pending_warm.empty() && small_path && warm_keys.size() < 64) warm_keys.push_back(pending_warm);
    pending_warm.clear();
  }
  hipStream_t st2 = nullptr;  // second stream, LOW priority: bandwidth-bound work beside the main stream (K_uf prefetch, colstats)
  hipStream_t st3 = nullptr;  // third stream, HIGH priority like the main one: the q(u)-only chains (S, S^-1; dL/dL, D2H)
  hipEvent_t ev_qu = nullptr;   // behind an in-place update of the resident q(u) (hmogp_qu_natgrad)
  hipEvent_t ev_fork = nullptr, ev_gsk = nullptr, ev_zero = nullptr, ev_info = nullptr, ev_S = nullptr, ev_join = nullptr, ev_col = nullptr, ev_kuf = nullptr, ev_params = nullptr,
             ev_ua = nullptr;

  hipEvent_t new_event() {
    if (pool_used == pool.size()) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      pool.push_back(e);
    }
    return pool[pool_used++];
  }
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
  void collect_spans() {
    static const bool dbg = getenv("HMOGP_DEBUG_TIMELINE") != nullptr;   // start / end of every span relative to the step's start
    for (auto& s : spans) {
      float f = 0.f;
      if (hipEventElapsedTime(&f, s.a, s.b) == hipSuccess) ms[s.cat] += f;
      if (dbg) {
        float t0 = 0.f;
        if (hipEventElapsedTime(&t0, ev_begin0, s.a) == hipSuccess)
          std::fprintf(stderr, "[hmogp timeline] cat %d  start %9.3f ms  dur %9.3f ms\n", s.cat, t0, f);
      }
    }
    spans.clear();
    pool_used = 0;
  }

  // ---- native exchange step (hmogp_comm_*): one RCCL communicator per engine, collectives on the engine's stream ----
  ncclComm_t comm = nullptr;
  int comm_ranks = 1, comm_rank = 0;
  bool exchanged = false;      // the bundle of the current step has been all-reduced

  void comm_init(int nranks, int rank, const void* id) {
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) throw EngineError{HMOGP_E_INVALID, "bad communicator arguments"};
    if (comm) throw EngineError{HMOGP_E_STATE, "this engine already has a communicator (hmogp_comm_destroy first)"};
    RcclApi& r = rccl();
    if (!r.ok()) throw EngineError{HMOGP_E_COMM, "librccl not available: " + r.why};
    HIP_TRY(hipSetDevice(device));
    hm_nccl::UniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    wire.ensure(sizeof(double) * nwire, true);   // allocated (and zeroed) before the first collective, outside any timing
    RCCL_TRY(r.commInitRank(&comm, nranks, uid, rank));
    comm_ranks = nranks, comm_rank = rank;
  }
  void comm_destroy() {
    if (!comm) return;
    (void)hipSetDevice(device);
    (void)hipStreamSynchronize(st);
    (void)rccl().commDestroy(comm);
    comm = nullptr, comm_ranks = 1, comm_rank = 0;
  }
  // A rank that cannot contribute to the step's collective (its row pass failed: HIP OOM, bad row range, E_STATE ...) ABORTS
  // the communicator, so that the peers' ncclAllReduce ends with an error instead of blocking for ever; the engine
  // is left without a communicator (hmogp_comm_info: 0 ranks) and every later sharded call fails with HMOGP_E_STATE.
  void comm_abort() {
    if (!comm) return;
    RcclApi& r = rccl();
    if (r.commAbort) (void)r.commAbort(comm);
    else (void)r.commDestroy(comm);
    comm = nullptr, comm_ranks = 1, comm_rank = 0;
  }
  // Wait for the engine's stream while a collective is in flight: the torch path this replaces has a watchdog, RCCL alone has
  // none.  Polls the stream, the communicator's asynchronous error state and a deadline (HMOGP_COMM_TIMEOUT_S, default 600 s;
  // 0 = wait for ever); on either failure the communicator is aborted and HMOGP_E_COMM is reported.
  void wait_exchanged() {
    static const double limit_s = [] {
      const char* e = getenv("HMOGP_COMM_TIMEOUT_S");
      return e ? atof(e) : 600.0;
    }();
    RcclApi& r = rccl();
    const auto t0 = std::chrono::steady_clock::now();
    for (long long spin = 0;; ++spin) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) throw HipError{q, "hipStreamQuery(st)", __FILE__, __LINE__};
      if ((spin & 1023) == 1023 && comm) {
        int async = hm_nccl::Success;
        if (r.commGetAsyncError && r.commGetAsyncError(comm, &async) == hm_nccl::Success && async != hm_nccl::Success &&
            async != hm_nccl::InProgress) {
          comm_abort();
          throw EngineError{HMOGP_E_COMM, std::string("RCCL: asynchronous error in the exchange step: ") + r.getErrorString(async)};
        }
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (limit_s > 0.0 && el > limit_s) {
          comm_abort();
          throw EngineError{HMOGP_E_COMM, "the exchange step did not complete within HMOGP_COMM_TIMEOUT_S (a peer rank is missing?): communicator aborted"};
        }
        if (el > 0.05) std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    }
  }
  // pack -> ncclAllReduce(sum, fp64, in place on the wire buffer) -> unpack, all ENQUEUED on the engine's stream: no host
  // synchronisation, no other library's stream.  The wire format holds the lower triangles of H_q only (12.7 MB instead
  // of 25.2 MB at M = 1024, Q = 3).
  void exchange() {
    if (!began) throw EngineError{HMOGP_E_STATE, "exchange outside hmogp_step_begin .. hmogp_step_finish"};
    if (!comm) throw EngineError{HMOGP_E_STATE, "no communicator (hmogp_comm_init)"};
    if (exchanged) throw EngineError{HMOGP_E_STATE, "the bundle of this step has already been exchanged"};
    HIP_TRY(hipSetDevice(device));
    Scope sc(this, CAT_EXCHANGE, 3);
    launch_wire_copy(stats.d(), wire.d(), NG, Q, M, per_q, 0, st);
    RCCL_TRY(rccl().allReduce(wire.p, wire.p, (size_t)nwire, hm_nccl::DataDouble, hm_nccl::OpSum, comm, st));
    launch_wire_copy(stats.d(), wire.d(), NG, Q, M, per_q, 1, st);
    exchanged = true;
  }

  ~hmogp_engine() {
    drop_graphs();
    comm_destroy();
    for (auto e : pool) (void)hipEventDestroy(e);
    if (h_info2) (void)hipHostFree(h_info2);
    if (ev_ng) (void)hipEventDestroy(ev_ng);
    if (h_small) (void)hipHostFree(h_small);
    for (auto e : {ev_begin0, ev_begin1, ev_fin0, ev_fin1, ev_fork, ev_gsk, ev_zero, ev_info, ev_S, ev_join, ev_col, ev_kuf, ev_params, ev_ua, ev_qu})
      if (e) (void)hipEventDestroy(e);
    if (hstage) (void)hipHostFree(hstage);
    if (h_info) (void)hipHostFree(h_info);
    if (st2_own) (void)hipStreamDestroy(st2_own);
    if (st3_own) (void)hipStreamDestroy(st3_own);
    if (st) (void)hipStreamDestroy(st);
  }

  double* Hq(int q) { return stats.d() + NG + q * per_q; }

  void init(const hmogp_config* c) {
    if (!c || c->abi_version != HMOGP_ABI_VERSION) throw EngineError{HMOGP_E_INVALID, "bad config / ABI version"};
    T = c->T, Q = c->Q, M = c->M, P = c->P, Df = c->Df, device = c->device;
    if (T < 1 || Q < 1 || M < 1 || Df < 1) throw EngineError{HMOGP_E_INVALID, "T, Q, M, Df must be >= 1"};
    if (P < 1 || P > 4) throw EngineError{HMOGP_E_INVALID, "input dimension P must be 1..4"};
    if (Q > HMOGP_MAXQ) throw EngineError{HMOGP_E_INVALID, "Q exceeds HMOGP_MAXQ (8)"};
    if (c->chunk_rows > 0) {
      chunk = c->chunk_rows;
    } else {  // default pool: up to 2^20 rows, the K^ / P~ workspaces (2 * Q * rows * M doubles) kept under ~64 GB of the 288
      const long long fit = (64LL << 30) / (16LL * Q * M);
      chunk = std::max<long long>(4096, std::min<long long>(1048576, fit / 1024 * 1024));
    }
    use_windows = (c->flags & HMOGP_CFG_EXACT_ZERO_WINDOWS) != 0;
    cache_kuu = (c->flags & HMOGP_CFG_CACHE_KUU) != 0;
    no_small = (c->flags & HMOGP_CFG_NO_SMALL_PATH) != 0;
    strict = strict_cfg = (c->flags & HMOGP_CFG_STRICT_QF) != 0;
    if (strict && use_windows) throw EngineError{HMOGP_E_INVALID, "HMOGP_CFG_STRICT_QF and HMOGP_CFG_EXACT_ZERO_WINDOWS exclude each other"};
    if (c->flags & ~(HMOGP_CFG_EXACT_ZERO_WINDOWS | HMOGP_CFG_CACHE_KUU | HMOGP_CFG_NO_SMALL_PATH | HMOGP_CFG_STRICT_QF))
      throw EngineError{HMOGP_E_INVALID, "unknown bits in hmogp_config.flags"};
    quirks = c->quirks;
    if (quirks & ~HMOGP_QUIRKS_REFERENCE) throw EngineError{HMOGP_E_INVALID, "unknown bits in hmogp_config.quirks"};
    if (use_windows && M > 8192) throw EngineError{HMOGP_E_INVALID, "exact-zero windows support M <= 8192"};
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
      throw EngineError{HMOGP_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)"};
    if (device < 0 || device >= ndev) throw EngineError{HMOGP_E_NO_DEVICE, "HIP device ordinal out of range"};
    HIP_TRY(hipSetDevice(device));
    {  // the main stream carries the latency-bound chains: highest priority, so that its (small) launches are dispatched
       // ahead of the bandwidth-bound work that runs beside them on the second stream
      int lo = 0, hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
      // HMOGP_ST2_CUS=<n> (experiment): give the second stream a CU mask of n of the device's CUs instead of a low
      // priority, so that the latency-bound chains of the other streams always find free CUs beside its HBM-bound work
      const char* cus_env = getenv("HMOGP_ST2_CUS");
      int cus = cus_env ? atoi(cus_env) : 0;
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, device));
      if (cus > 0 && cus < prop.multiProcessorCount) {
        std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
        for (int i = 0; i < cus; ++i) mask[i / 32] |= 1u << (i % 32);
        if (hipExtStreamCreateWithCUMask(&st2, (uint32_t)mask.size(), mask.data()) != hipSuccess) st2 = nullptr;
      }
      // HMOGP_ST2_FREE=<n> with HMOGP_ST2_LAYOUT=0|1 (experiment): leave n CUs of EVERY XCD out of the second stream's mask --
      // a mask that drops whole XCDs unbalances kernels whose blocks are dealt round-robin over the XCDs.  Layout 0: mask bit
      // i = CU i % 32 of XCD i / 32; layout 1: bit i = CU i / 8 of XCD i % 8.
      const char* free_env = getenv("HMOGP_ST2_FREE");
      const int nfree = free_env ? atoi(free_env) : 0;
      if (!st2 && nfree > 0 && nfree < 32 && prop.multiProcessorCount == 256) {
        const char* lay = getenv("HMOGP_ST2_LAYOUT");
        const int layout = lay ? atoi(lay) : 1;
        st2_masked = true;
        std::vector<uint32_t> mask(8, 0xFFFFFFFFu);
        for (int x = 0; x < 8; ++x)
          for (int c = 32 - nfree; c < 32; ++c) {
            const int bit = layout == 0 ? x * 32 + c : c * 8 + x;
            mask[bit / 32] &= ~(1u << (bit % 32));
          }
        if (hipExtStreamCreateWithCUMask(&st2, (uint32_t)mask.size(), mask.data()) != hipSuccess) st2 = nullptr, st2_masked = false;
      }
      if (!st2) HIP_TRY(hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, lo));
      HIP_TRY(hipStreamCreateWithPriority(&st3, hipStreamNonBlocking, hi));
      st2_own = st2, st3_own = st3;
    }
    for (hipEvent_t* e : {&ev_fork, &ev_gsk, &ev_zero, &ev_info, &ev_S, &ev_join, &ev_col, &ev_kuf, &ev_params, &ev_ua, &ev_qu})
      HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev_qu, st));
    HIP_TRY(hipEventRecord(ev_join, st));
    for (hipEvent_t* e : {&ev_begin0, &ev_begin1, &ev_fin0, &ev_fin1}) HIP_TRY(hipEventCreate(e));
    f_index.assign(c->f_index, c->f_index + Df);
    d_index.assign(c->d_index, c->d_index + Df);
    tasks.resize(T);
    int d = 0;
    for (int t = 0; t < T; ++t) {
      Task& k = tasks[t];
      k.lik = c->lik_id[t];
      k.param = c->lik_param ? c->lik_param[t] : 0.0;
      if (k.lik == HMOGP_LIK_GAUSSIAN && !(k.param > 0.0)) k.param = 0.5;  // gaussian.py:21-24
      k.dimf = lik_dimf(k.lik, k.param);
      if (k.dimf < 1 || k.dimf > HMOGP_MAXJ) throw EngineError{HMOGP_E_INVALID, "unsupported likelihood / dim_f"};
      k.d0 = d;
      for (int j = 0; j < k.dimf; ++j, ++d)
        if (d >= Df || f_index[d] != t || d_index[d] != j)
          throw EngineError{HMOGP_E_INVALID, "f_index / d_index inconsistent with the likelihood list"};
    }
    if (d != Df) throw EngineError{HMOGP_E_INVALID, "Df does not match the likelihood list"};
    // bundle layout
    const long long MM = (long long)M * M;
    NG = 2 + Df;
    oR = MM, oDZ = MM + M, oSA = oDZ + (long long)M * P, oSL = oSA + 1, oSWK = oSA + 2;
    per_q = oSWK + Df;
    nstats = NG + Q * per_q;
    stats.ensure(sizeof(double) * nstats, true);
    nwire = NG + Q * ((long long)M * (M + 1) / 2 + (per_q - MM));
    for (int t = 0; t < T; ++t) {
      Task& k = tasks[t];
      const int J = k.dimf;
      k.nscal = 2 + 2 * Q + J + Q * J;
      std::vector<long long> off(k.nscal);
      off[0] = 0, off[1] = 1;
      for (int q = 0; q < Q; ++q) {
        off[2 + 2 * q] = NG + q * per_q + oSA;
        off[3 + 2 * q] = NG + q * per_q + oSL;
        for (int j = 0; j < J; ++j) off[2 + 2 * Q + J + q * J + j] = NG + q * per_q + oSWK + k.d0 + j;
      }
      for (int j = 0; j < J; ++j) off[2 + 2 * Q + j] = 2 + k.d0 + j;
      k.offsets.ensure(sizeof(long long) * k.nscal);
      HIP_TRY(hipMemcpy(k.offsets.p, off.data(), sizeof(long long) * k.nscal, hipMemcpyHostToDevice));
    }
    // parameter + M x M buffers
    const size_t mmq = sizeof(double) * MM * Q;
    for (DevBuf* b : {&Kuu, &Luu, &Kuui, &L, &S, &KiS, &KSK, &C, &Ctri, &Sqi, &tmpA, &tmpB, &HK, &G, &GSK, &dKmm, &dLdS}) b->ensure(mmq, true);
    // ALL parameters live in ONE device block [ hypers + jitter | Z | m_u | L_flat ] (segments 16-byte aligned): large models fill
    // the segments by separate copies straight from the caller's arrays, small-problem mode by ONE copy from a page-locked image
    // (a host-bound small-model step pays ~4-8 us of API time and ~4 us of device time per hipMemcpyAsync)
    // variance | lengthscale | W | kappa | jitter of the small path | chain-factor W0 (quirk Q3) | batch scales | evaluation counter
    oJit = 2 * Q + 2 * Q * Df, oW0 = oJit + Q, oBs = oW0 + Q * Df, oSeq = oBs + T;
    n_small = oSeq + 1;
    auto even = [](long long n) { return (n + 1) & ~1LL; };
    const long long nZ = (long long)M * Q * P, nmu = (long long)M * Q, nL = ((long long)M * (M + 1) / 2) * Q;
    oZ = even(n_small), oMu = oZ + even(nZ), oLf = oMu + even(nmu), n_params = oLf + even(nL);
    dparams.ensure(sizeof(double) * n_params);
    dsmall.view(dparams.d(), sizeof(double) * n_small);
    dZ.view(dparams.d() + oZ, sizeof(double) * nZ), dmu.view(dparams.d() + oMu, sizeof(double) * nmu);
    dLflat.view(dparams.d() + oLf, sizeof(double) * nL);
    HIP_TRY(hipHostMalloc((void**)&h_small, sizeof(double) * (M <= 128 ? n_params : n_small), hipHostMallocDefault));
    dvar.view(dsmall.d(), sizeof(double) * Q), dell.view(dsmall.d() + Q, sizeof(double) * Q);
    dW.view(dsmall.d() + 2 * Q, sizeof(double) * Q * Df), dkap.view(dsmall.d() + 2 * Q + Q * Df, sizeof(double) * Q * Df);
    a.ensure(sizeof(double) * Q * M), Kr.ensure(sizeof(double) * Q * M), gmu.ensure(sizeof(double) * Q * M);
    gL.ensure(sizeof(double) * ((long long)M * (M + 1) / 2) * Q);
    klout.ensure(sizeof(double) * Q * KL_BLOCKS * 6, true);   // KL partials [Q][KL_BLOCKS][5] | diag(K_uu^-1) block maxima [Q][KL_BLOCKS]
    rowout.ensure(sizeof(double) * Q * M * (2 + P));
    dinfo.ensure(sizeof(int) * (2 * HMOGP_MAXQ + 2), true), djit.ensure(sizeof(double) * Q), dscr.ensure(sizeof(double) * Q * M * M);
    rung.assign(Q, -1);
  }

  void set_task_data(int t, const double* X, const double* Y, long long N) {
    if (t < 0 || t >= T || N < 0 || (N > 0 && (!X || !Y))) throw EngineError{HMOGP_E_INVALID, "bad task data"};
    Task& k = tasks[t];
    k.N = N;
    began = false;
    staged_key.clear();
    drop_graphs();
    if (N == 0) return;
    k.X.ensure(sizeof(double) * N * P);
    k.Y.ensure(sizeof(double) * N);
    HIP_TRY(hipMemcpy(k.X.p, X, sizeof(double) * N * P, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(k.Y.p, Y, sizeof(double) * N, hipMemcpyHostToDevice));
    if (k.lik == HMOGP_LIK_POISSON) {  // gammaln(y+1) depends on the data only (poisson.py:33)
      k.Yaux.ensure(sizeof(double) * N);
      launch_gammaln1p(k.Y.d(), k.Yaux.d(), N, st);
      HIP_TRY(hipStreamSynchronize(st));
    }
  }

  void ensure_workspace(long long rows) {
    rows = std::max<long long>(rows, 1);
    if (rows <= ws_rows) return;
    const size_t nm = sizeof(double) * rows * M * Q, nv = sizeof(double) * rows * Q;
    Kh.ensure(nm), Pt.ensure(nm), Xws.ensure(sizeof(double) * rows * P);
    staged_key.clear();
    drop_graphs(true);   // (the evaluation that grows the workspaces runs normally to its end: its key stays warm)
    for (DevBuf* b : {&vp, &vc, &vpt, &vct, &valpha, &vbeta, &valpha0, &vbeta0}) b->ensure(nv, true);
    ws_strict_rows = 0;          // (the strict mode's extra workspaces follow lazily: ensure_strict_workspace)
    colpart.ensure(sizeof(double) * std::max((rows + 255) / 256, (std::min<long long>(rows, 16384) + 31) / 32) * M * (2 + P) * Q);
    colred.ensure(sizeof(double) * M * Q);
    quadpart.ensure(sizeof(double) * (rows * 64 / 256 + 1 + HMOGP_QUAD_MULTI) * HMOGP_MAXSCAL);
    fwdpart.ensure(sizeof(double) * 4 * FWD_PARTS * ((M + 127) / 128) * rows * Q);  // 4 statistics x FWD_PARTS wave columns per tile
    if (use_windows) {
      const size_t tiles = (rows + 127) / 128, ncb = (M + 127) / 128;
      winrow.ensure(sizeof(int) * 2 * tiles * Q), wincol.ensure(sizeof(int) * 2 * ncb * Q), winhit.ensure(tiles * ncb);
    }
    ws_rows = rows;
  }

  // the strict mode's own buffers, allocated when an evaluation first runs in that mode (config flag or per-evaluation flag)
  long long ws_strict_rows = 0;
  void ensure_strict_workspace() {
    if (!strict) return;
    Dm.ensure(sizeof(double) * (long long)M * M * Q, true);
    if (ws_strict_rows >= ws_rows) return;
    Ah.ensure(sizeof(double) * ws_rows * M * Q);
    vpg.ensure(sizeof(double) * ws_rows * Q, true), vcg.ensure(sizeof(double) * ws_rows * Q, true);
    ws_strict_rows = ws_rows;
  }

  // ------------------------------------------------------------------------------------------ parameters
  void upload_params(const hmogp_params* p, bool enqueue = true) {
    if (!p || !p->Z || !p->variance || !p->lengthscale || !p->W || !p->kappa)
      throw EngineError{HMOGP_E_INVALID, "missing parameter array"};
    const bool resident = !p->m_u && !p->L_flat;     // q(u) stays where hmogp_qu_load / hmogp_qu_adadelta left it
    if (resident ? !qu_resident : (!p->m_u || !p->L_flat))
      throw EngineError{HMOGP_E_INVALID, resident ? "m_u / L_flat are NULL but no q(u) is resident (hmogp_qu_load)" : "missing parameter array"};
    if (!resident) qu_resident = false;              // host arrays overwrite the resident copy
    const long long Mtri = (long long)M * (M + 1) / 2;
    h_var.assign(p->variance, p->variance + Q);
    h_ell.assign(p->lengthscale, p->lengthscale + Q);
    h_W.assign(p->W, p->W + Q * Df);
    h_kap.assign(p->kappa, p->kappa + Q * Df);
    const bool stale = (quirks & HMOGP_QUIRK_STALE_W) != 0;   // exact mode: the chain factors are the live W / kappa
    const double* w0 = (stale && p->W0) ? p->W0 : p->W;
    const double* k0 = (stale && p->kappa0) ? p->kappa0 : p->kappa;
    h_W0.assign(w0, w0 + Q * Df);
    h_kap0.assign(k0, k0 + Q * Df);
    h_bs.assign(T, 1.0);
    if (p->batch_scale) h_bs.assign(p->batch_scale, p->batch_scale + T);
    rb.assign(T, 0), re.resize(T);
    for (int t = 0; t < T; ++t) {
      re[t] = tasks[t].N;
      if (p->row_begin) rb[t] = p->row_begin[t];
      if (p->row_end) re[t] = p->row_end[t];
      if (rb[t] < 0 || re[t] > tasks[t].N || rb[t] > re[t]) throw EngineError{HMOGP_E_INVALID, "row range outside the task's data"};
    }
    h_Z.assign(p->Z, p->Z + (size_t)M * Q * P);
    rung_request.resize(Q);
    for (int q = 0; q < Q; ++q) {
      rung[q] = p->forced_rung ? p->forced_rung[q] : -2;
      rung_request[q] = rung[q];
      if (!(h_ell[q] > 0.0)) throw EngineError{HMOGP_E_INVALID, "lengthscale must be positive"};
    }
    group_mask = p->group_mask;
    // (h_small is re-written only after the previous evaluation has synchronised the stream that read it)
    std::copy(h_var.begin(), h_var.end(), h_small);
    std::copy(h_ell.begin(), h_ell.end(), h_small + Q);
    std::copy(h_W.begin(), h_W.end(), h_small + 2 * Q);
    std::copy(h_kap.begin(), h_kap.end(), h_small + 2 * Q + Q * Df);
    for (int q = 0; q < Q; ++q)      // small path: jitter of a forced rung (GPy jitchol: mean(diag) 1e-6 10^k, diag(K_uu) = variance)
      h_small[oJit + q] = rung[q] >= 0 ? h_var[q] * 1e-6 * std::pow(10.0, rung[q]) : 0.0;
    std::copy(h_W0.begin(), h_W0.end(), h_small + oW0);
    std::copy(h_bs.begin(), h_bs.end(), h_small + oBs);
    eval_seq = eval_seq >= (1 << 30) ? 1 : eval_seq + 1;
    h_small[oSeq] = (double)eval_seq;
    if (!enqueue) {                   // replay of a captured graph: the page-locked image is all the graph's upload node reads
      std::memcpy(h_small + oZ, p->Z, sizeof(double) * M * Q * P);
      if (!resident) {
        std::memcpy(h_small + oMu, p->m_u, sizeof(double) * M * Q);
        std::memcpy(h_small + oLf, p->L_flat, sizeof(double) * Mtri * Q);
      }
      return;
    }
    // (in-place updates of the resident q(u) -- Adadelta, natural gradient -- run on the main stream without a host
    //  synchronisation: whatever touches q(u) on the third stream next is ordered behind them; one stream in small-problem mode)
    if (!small_mode) HIP_TRY(hipStreamWaitEvent(st3, ev_qu, 0));
    if (small_mode && M <= 128) {     // one image, one copy: [ hypers | Z | (m_u | L_flat unless q(u) is resident) ]
      std::memcpy(h_small + oZ, p->Z, sizeof(double) * M * Q * P);
      long long n_up = oMu;
      if (!resident) {
        std::memcpy(h_small + oMu, p->m_u, sizeof(double) * M * Q);
        std::memcpy(h_small + oLf, p->L_flat, sizeof(double) * Mtri * Q);
        n_up = n_params;
      }
      HIP_TRY(hipMemcpyAsync(dparams.p, h_small, sizeof(double) * n_up, hipMemcpyHostToDevice, st));
    } else {
      HIP_TRY(hipMemcpyAsync(dZ.p, p->Z, sizeof(double) * M * Q * P, hipMemcpyHostToDevice, st));
      if (!resident) HIP_TRY(hipMemcpyAsync(dmu.p, p->m_u, sizeof(double) * M * Q, hipMemcpyHostToDevice, st));
      // the one large parameter (12.6 MB at M = 1024, Q = 3) goes up on the second stream, whose chain is its only consumer
      // (u_algebra): the K_uu chain on the main stream starts without waiting for it
      if (!resident) HIP_TRY(hipMemcpyAsync(dLflat.p, p->L_flat, sizeof(double) * Mtri * Q, hipMemcpyHostToDevice, st3));
      HIP_TRY(hipMemcpyAsync(dsmall.p, h_small, sizeof(double) * n_small, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipEventRecord(ev_params, st));   // what the second stream has to wait for before it reads Z / the hypers
  }

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
  void u_algebra_small() {
    Scope sc(this, CAT_MM, 1);
    kuu_key_valid = false;
    if (!h_info) HIP_TRY(hipHostMalloc((void**)&h_info, sizeof(int) * HMOGP_MAXQ, hipHostMallocDefault));
    // (the jitter of a forced rung went up with the hyper-parameter block: upload_params)
    for (int q = 0; q < Q; ++q)
      if (rung[q] == -2) rung[q] = -1;
    // (no memsets: u_small_kernel always writes the info words, compares its hand-over flags with the evaluation counter of the
    //  parameter block, and zeroes the statistic bundle the row pass accumulates into)
    if (!pools.empty()) {
      stage_pool_inputs(pools[0], st);
      if (!small_rows) kuf_pool(pools[0], st);      // (the fused forward kernel builds K^ itself)
      kuf_prefetched = true;
      HIP_TRY(hipEventRecord(ev_kuf, st));
    }
    SmallU u;
    u.M = M, u.Q = Q, u.P = P, u.ldz = Q * P;
    u.Z = dZ.d(), u.var = dvar.d(), u.ell = dell.d(), u.jit = dsmall.d() + oJit, u.mu = dmu.d(), u.Lflat = dLflat.d();
    u.Kuu = Kuu.d(), u.Luu = Luu.d(), u.Kuui = Kuui.d(), u.L = L.d(), u.S = S.d(), u.KiS = KiS.d(), u.KSK = KSK.d(), u.C = C.d();
    u.Ctri = Ctri.d(), u.Sqi = Sqi.d(), u.a = a.d(), u.klout = klout.d(), u.info = dinfo.as<int>(), u.flag = dinfo.as<int>() + HMOGP_MAXQ;
    u.seq = dsmall.d() + oSeq;
    u.zero = stats.d(), u.nzero = nstats;   // (also without rows: hmogp_step_finish reads the bundle)
    launch_u_small(u, st);
    // (the info words reach the host with the results of hmogp_step_finish -- its last block gathers them -- unless the caller
    //  needs them behind hmogp_step_begin already)
    if (info_early) HIP_TRY(hipMemcpyAsync(h_info, dinfo.p, sizeof(int) * Q, hipMemcpyDeviceToHost, st));
    small_info_pending = true;
    HIP_TRY(hipEventRecord(ev_join, st));    // (what hmogp_step_finish orders itself behind on the regular path)
  }

  void u_algebra() {
    if (small_path) return u_algebra_small();
    Scope sc(this, CAT_MM, 0);
    const long long MM = (long long)M * M;
    const int ldz = Q * P;
    // K_uu, its jittered Cholesky factor and inverse depend on (Z, variance, lengthscale, forced rungs) only.  With
    // HMOGP_CFG_CACHE_KUU they are reused while those inputs are bit-identical to the previous evaluation's -- the
    // variational E-steps of VEM / SVI change q(u) only (util.py:294-306, svmogp.py:188-199).  The reference recomputes
    // them on every call (util.py:181-200); the result is the same.
    // The chain that only depends on q(u)'s factor -- L, S = L L^T, S^-1 -- runs on a second stream, concurrently with
    // the (latency-bound, few-CU) factorisation and inversion of K_uu; scratch: HK, G (unused before hmogp_step_finish).
    // Launch order on the host = critical path first: the K_uu chain (covariance, 32 dependent factorisation launches) is
    // enqueued before anything else, so that the device starts on it while the host is still enqueueing the q(u) chain and
    // the K_uf prefetch on the second stream (enqueued the other way round, the chain used to start ~0.35 ms late).
    std::vector<double> key;
    if (cache_kuu) {
      key.assign(h_Z.begin(), h_Z.end());
      key.insert(key.end(), h_var.begin(), h_var.end());
      key.insert(key.end(), h_ell.begin(), h_ell.end());
      for (int q = 0; q < Q; ++q) key.push_back((double)rung_request[q]);
      key.push_back(strict ? 1.0 : 0.0);     // (the strict mode forms K_uu^-1 by substitution: not interchangeable)
    }
    const bool kuu_hit = cache_kuu && kuu_key_valid && key.size() == kuu_key.size() &&
                         std::memcmp(key.data(), kuu_key.data(), sizeof(double) * key.size()) == 0;
    JitcholState js;
    if (!h_info) HIP_TRY(hipHostMalloc((void**)&h_info, sizeof(int) * HMOGP_MAXQ, hipHostMallocDefault));
    js.info = h_info;
    if (kuu_hit) {
      rung = kuu_rung;
    } else {
      kuu_key_valid = false;
      RbfBatch kb;  // K_uu of all latents in one launch; both arguments passed (util.py:197) -> no forced diagonal
      kb.nq = Q, kb.var = dvar.d(), kb.ell = dell.d(), kb.sZ = P, kb.sX = P, kb.sK = MM;
      launch_rbf(dZ.d(), ldz, M, P, dZ.d(), ldz, M, 0.0, 1.0, Kuu.d(), false, st, nullptr, true, &kb);
      jitchol_enqueue(Kuu.d(), Luu.d(), Q, M, h_var.data(), rung.data(), dinfo.as<int>(), djit.d(), dscr.d(), st, js, 0);
    }
    // (only the first panels of the factorisation are enqueued at this point -- enough device work to cover the host time
    // of the launches below; the rest follows them)
    // K_uf of the first pool only needs X, Z and the kernel hyper-parameters: it is built on the low-priority second
    // stream beside the latency-bound chains.  Its exp() work and the matrix cores share the FP64 pipe (tools/probes/
    // probe_coissue.hip), so hiding it behind the forward contraction gains nothing -- the chains, which need neither,
    // are the one place where it is free.
    HIP_TRY(hipStreamWaitEvent(st2, ev_params, 0));
    if (!pools.empty()) {
      // (with it, off the critical path of the main stream: the zeroed statistic bundle and the pool-contiguous inputs)
      HIP_TRY(hipMemsetAsync(stats.p, 0, sizeof(double) * nstats, st2));
      stage_pool_inputs(pools[0], st2);
      kuf_pool(pools[0], st2);
      HIP_TRY(hipEventRecord(ev_kuf, st2));
      kuf_prefetched = true;
    }
    // The q(u) chain goes to a stream of the SAME (high) priority as the main one: on the low-priority stream it would
    // not be dispatched before the 32 back-to-back factorisation launches of the main stream have drained.
    if (!kuu_hit) {   // the zeroed target of the K_uu chain's triangular inverse: 25 MB memset, not on the chain's stream
      HIP_TRY(hipMemsetAsync(tmpA.p, 0, sizeof(double) * MM * Q, st3));
      HIP_TRY(hipEventRecord(ev_zero, st3));
    }
    launch_unpack_tril(dLflat.d(), L.d(), Q, M, st3);             // flat_to_triang   (svmogp_inf.py:193)
    mm(L.d(), false, L.d(), false, S.d(), 1.0, -1, -1, st3, +1, -1);  // S = L L^T    (:194-195), L lower
    HIP_TRY(hipEventRecord(ev_S, st3));
    launch_trtri_batched(L.d(), HK.d(), G.d(), Q, M, st3);        // S^-1 = dpotri(L) (svmogp_inf.py:124)
    launch_ltl_batched(HK.d(), Sqi.d(), Q, M, st3);
    HIP_TRY(hipEventRecord(ev_join, st3));
    if (!kuu_hit && !js.complete)
      jitchol_enqueue(Kuu.d(), Luu.d(), Q, M, h_var.data(), rung.data(), dinfo.as<int>(), djit.d(), dscr.d(), st, js, 1);
    if (!kuu_hit) HIP_TRY(hipEventRecord(ev_info, st));          // behind the read-back of the factorisation's info
    // Everything behind the factorisation is enqueued SPECULATIVELY, before the host knows whether it succeeded: the device
    // goes straight on while the host waits for `info` alone (an event, not the stream) and then enqueues the row pass
    // behind ~0.6 ms of queued work -- no host round trip in the latency-bound chain.  If a latent did fail (GPy's jitter
    // ladder is needed: rare), the ladder runs synchronously as before and the same launches are simply issued again.
    auto tail = [&](bool first) {
      if (!kuu_hit && strict) {
        // strict mode: K_uu^-1 = dpotrs(Luu, I) by the blocked substitution, lower triangle mirrored like GPy's dpotri wrapper
        // (util.py:199).  The merge-based triangular inverse below is ~100x further from LAPACK's dpotri where it matters here
        // (|K_uu^-1 K_uu - I| 8.6e-8 against 3e-10 at cond 1e7) -- invisible at cond <= 1e5, 2e-8 of g_W / g_Z at 1e7.
        if (first) HIP_TRY(hipStreamWaitEvent(st, ev_zero, 0));
        launch_identity(Kuui.d(), Q, M, st);
        potrs_rows_inplace(Kuui.d(), MM, Luu.d(), MM, M, M, Q, st);
        launch_mirror_lower(Kuui.d(), Q, M, MM, st);
      } else if (!kuu_hit) {
        if (first) HIP_TRY(hipStreamWaitEvent(st, ev_zero, 0));     // (tmpA zeroed on the third stream, above)
        launch_trtri_batched(Luu.d(), tmpA.d(), tmpB.d(), Q, M, st, first);
        launch_ltl_batched(tmpA.d(), Kuui.d(), Q, M, st);           // K_uu^-1          (util.py:199)
      }
      launch_gemv_batched(Kuui.d(), dmu.d(), a.d(), Q, M, 1, Q, st);  // a = K_uu^-1 m
      HIP_TRY(hipStreamWaitEvent(st, ev_S, 0));
      mm(Kuui.d(), false, S.d(), true, KiS.d());
      if (strict) launch_strict_d(KiS.d(), Dm.d(), Q, M, st);        // S K_uu^-1 - I   (svmogp_inf.py:157-158)
      mm(KiS.d(), false, Kuui.d(), true, KSK.d());
      launch_sub(KSK.d(), Kuui.d(), C.d(), MM * Q, st);             // C = K^-1 S K^-1 - K^-1
      launch_tri_fold(C.d(), Ctri.d(), Q, M, st);                   // x^T Ctri x == x^T C x with a triangular matrix
      // (the main stream does NOT wait for S^-1 here: the row pass needs C only, S^-1 is consumed on the third stream --
      // KL terms, dL/dS -- and hmogp_step_finish orders itself behind that chain before it reuses its scratch buffers.
      // With a cached K_uu chain this wait used to hold the forward contraction back by ~0.3 ms.)
      // the KL terms (svmogp_inf.py:227-250) only need what exists now: they run on the third stream beside the row pass
      // instead of sitting in the tail of hmogp_step_finish
      HIP_TRY(hipEventRecord(ev_ua, st));
      HIP_TRY(hipStreamWaitEvent(st3, ev_ua, 0));
      launch_kl_terms(Kuui.d(), S.d(), dmu.d(), a.d(), Luu.d(), L.d(), Sqi.d(), Q, M, klout.d(), st3);
    };
    tail(true);
    if (!kuu_hit) {
      HIP_TRY(hipEventSynchronize(ev_info));
      bool failed = false;
      for (int q = 0; q < Q; ++q) failed = failed || js.info[q] != 0;
      if (failed) {
        jitchol_resolve(Kuu.d(), Luu.d(), Q, M, h_var.data(), rung.data(), dinfo.as<int>(), djit.d(), dscr.d(), st, js);
        tail(false);
      }
      if (cache_kuu) kuu_key.swap(key), kuu_rung = rung, kuu_key_valid = true;
    }
  }

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
  void plan_pools() {
    pools.clear();
    kuf_prefetched = false;
    std::vector<Seg> cur;
    long long fill = 0;
    for (int t = 0; t < T; ++t)
      for (long long r0 = rb[t]; r0 < re[t];) {
        const long long n = std::min(chunk - fill, re[t] - r0);
        cur.push_back(Seg{t, r0, n, fill});
        fill += n, r0 += n;
        if (fill == chunk || use_windows) pools.push_back(cur), cur.clear(), fill = 0;
      }
    if (!cur.empty()) pools.push_back(cur);
    long long maxrows = 1;
    for (auto& pl : pools) maxrows = std::max(maxrows, pl.back().off + pl.back().n);
    ensure_workspace(maxrows);
  }
  // K_uf = k_q(X, Z_q) of one pool, all latents in one launch per segment (grid.z = latent), on `stream`
  void kuf_pool(const std::vector<Seg>& pl, hipStream_t stream, size_t seg_begin = 0, size_t seg_end = (size_t)-1) {
    const int ldz = Q * P, ncb = (M + 127) / 128;
    const long long wtiles = (ws_rows + 127) / 128, sK = ws_rows * M;
    int* rw = use_windows ? winrow.as<int>() : nullptr;    // [Q][wtiles][2]
    int* cw = use_windows ? wincol.as<int>() : nullptr;    // [Q][ncb][2]
    seg_end = std::min(seg_end, pl.size());
    if (seg_begin >= seg_end) return;
    if (small_mode && pl.size() > 1 && seg_begin == 0 && seg_end == pl.size() && !use_windows) {
      // small-problem mode: the pool's rows are contiguous in Xws (stage_pool_inputs): ONE launch for all tasks and latents
      Scope sc(this, CAT_RBF, 1, stream);
      RbfBatch rbt;
      rbt.nq = Q, rbt.var = dvar.d(), rbt.ell = dell.d(), rbt.sZ = P, rbt.sK = sK, rbt.sWin = 2 * wtiles;
      const long long n = pl.back().off + pl.back().n;
      launch_rbf(Xws.d(), P, n, P, dZ.d(), ldz, M, 0.0, 1.0, Kh.d(), false, stream, nullptr, strict, &rbt);
      return;
    }
    Scope sc(this, CAT_RBF, (int)(seg_end - seg_begin) + (use_windows ? 3 * Q : 0), stream);
    for (size_t si = seg_begin; si < seg_end; ++si) {
      const Seg& sg = pl[si];
      const double* Xs = tasks[sg.t].X.d() + sg.r0 * P;
      if (use_windows)
        for (int q = 0; q < Q; ++q)
          launch_windows(Xs, sg.n, P, dZ.d() + q * P, ldz, M, h_ell[q], rw + 2 * wtiles * q, cw + 2 * ncb * q,
                         winhit.as<unsigned char>(), stream);
      RbfBatch rbt;
      rbt.nq = Q, rbt.var = dvar.d(), rbt.ell = dell.d(), rbt.sZ = P, rbt.sK = sK, rbt.sWin = 2 * wtiles;
      // On the side stream the construction is cut into launches of KUF_CHUNK_ROWS rows (~70 us each): a kernel that fills
      // every CU for a millisecond stalls every launch of the latency-bound chains on the other streams until it has
      // drained (stream priorities notwithstanding); between short launches they slip in.
      static const long long chunk_env = [] {   // HMOGP_KUF_CHUNK=<rows per launch on the side stream> (experiment)
        const char* e = getenv("HMOGP_KUF_CHUNK");
        return e ? atoll(e) : 0LL;
      }();
      const long long step = (stream != st && !use_windows) ? (chunk_env > 0 ? chunk_env : (st2_masked ? 100000LL : KUF_CHUNK_ROWS)) : sg.n;
      for (long long r = 0; r < sg.n; r += step)
        launch_rbf(Xs + r * P, P, std::min(step, sg.n - r), P, dZ.d(), ldz, M, 0.0, 1.0, Kh.d() + (sg.off + r) * M, false, stream,
                   rw, strict, &rbt);   // (strict q(f): GPy's rounding order, sqrt and divide included)
    }
  }

  // inputs of a multi-segment pool, contiguous in pool order (fs_x of the forward epilogue, the column statistics)
  // (the staged copy is reused while the SAME segments of the SAME data are asked for again -- every full-batch evaluation after
  //  the first: one D2D copy per task less per step, which matters for host-bound small models)
  std::vector<long long> staged_key;
  void stage_pool_inputs(const std::vector<Seg>& pl, hipStream_t stream) {
    if (pl.size() <= 1) return;
    std::vector<long long> key;
    for (auto& sg : pl) key.push_back(sg.t), key.push_back(sg.r0), key.push_back(sg.n), key.push_back(sg.off);
    if (pools.size() == 1 && key == staged_key) return;
    staged_key = pools.size() == 1 ? key : std::vector<long long>();
    for (auto& sg : pl)
      HIP_TRY(hipMemcpyAsync(Xws.d() + sg.off * P, tasks[sg.t].X.d() + sg.r0 * P, sizeof(double) * sg.n * P,
                             hipMemcpyDeviceToDevice, stream));
  }

  // strict q(f): the solve-based forms of svmogp_inf.py:212-218 for the n pool rows whose K^ sits in Kh (all latents batched).
  // A = K^ Kuu^-1 = dpotrs(Luu, K^T)^T by two BLOCKED TRIANGULAR SOLVES against Luu: 32-column diagonal blocks by true substitution
  // (trsm_diag_kernel), the updates between them as GEMMs -- backward stable like LAPACK's dtrsm.  (Round 5 first used two
  // products with the explicit Luu^-1: m_fd was then 9e-8 of its scale away from the reference at cond(K_uu) = 1e7, 1.4e-2 at
  // cond 1e12; with the substitution 3e-10 / the reference's own rounding sensitivity.)
  void strict_forward(long long n, const double* X, bool grads, bool hyper) {
    const long long MM = (long long)M * M, ldn = ws_rows, sK = ldn * M;
    Scope sc(this, CAT_FWD, 4 * ((M + 31) / 32) + (grads ? 4 : 2));
    auto rows_gemm = [&](const double* A_, const double* B_, int b_kmajor, int b_tri, double* C_) {
      GemmArgs g;
      g.A = A_, g.lda = M, g.a_kmajor = 0, g.sA = sK;
      g.B = B_, g.ldb = M, g.b_kmajor = b_kmajor, g.sB = MM, g.b_tri = b_tri;
      g.C = C_, g.ldc = M, g.sC = sK;
      g.M = (int)n, g.N = M, g.K = M;
      g.nbatch = Q;
      g.role = 1;                       // (no fused statistics: fs_part stays null) the specialised 8-wave forward kernel where the
      launch_gemm_rowpass_or_general(g, st);   // shape allows it -- incl. its triangular-fold pairing for T = A L_q -- else the general one
    };
    potrs_rows_inplace(Ah.d(), sK, Luu.d(), MM, M, n, Q, st, tmpB.d(), Kh.d());   // A = dpotrs(Luu, K^T)^T   (svmogp_inf.py:214-215; tmpB: free here)
    // T = A L_q = dtrmm(L_q^T, R)^T (:217) is only ever consumed as rowsum(T .* T) (:218): where the specialised fold kernel takes
    // the product, its epilogue forms that sum from the accumulators and T is neither written nor read back (2 x 19.7 GB at H)
    bool t2_fused = false;
    {
      GemmArgs g;
      g.A = Ah.d(), g.lda = M, g.a_kmajor = 0, g.sA = sK;
      g.B = L.d(), g.ldb = M, g.b_kmajor = 1, g.sB = MM, g.b_tri = +1;
      g.C = Pt.d(), g.ldc = M, g.sC = sK;
      g.M = (int)n, g.N = M, g.K = M;
      g.nbatch = Q;
      g.role = 1;
      const int tiles = (M + 127) / 128;
      g.fs_part = fwdpart.d(), g.fs_sPart = 4LL * FWD_PARTS * tiles * ldn, g.fs_sq = 1, g.store_c = 0;
      static const bool t2_env = [] {   // HMOGP_STRICT_T2=0: T stored and squared by strict_rowstats_kernel (A/B runs)
        const char* e = getenv("HMOGP_STRICT_T2");
        return !(e && e[0] == '0');
      }();
      if (t2_env && gemm_rowpass_would_take(g)) {
        const int nparts = launch_gemm_rowpass_or_general(g, st);
        launch_combine_parts(fwdpart.d(), nparts * tiles, n, nullptr, vct.d(), nullptr, nullptr, st, Q, g.fs_sPart, ldn);
        t2_fused = true;
      } else {
        rows_gemm(Ah.d(), L.d(), 1, +1, Pt.d());
      }
    }
    StrictRows sr;
    sr.M = M, sr.Q = Q, sr.P = P, sr.ldz = Q * P, sr.n = n, sr.ldn = ldn, sr.sK = sK, sr.sZ = P;
    sr.Kh = Kh.d(), sr.Ah = Ah.d(), sr.Tt = Pt.d(), sr.Pt = Pt.d(), sr.mu = dmu.d(), sr.a = a.d();
    sr.X = X, sr.Z = dZ.d(), sr.ell = dell.d();
    sr.p = vp.d(), sr.c = vc.d(), sr.pg = vpg.d(), sr.cg = vcg.d(), sr.pt = hyper ? vpt.d() : nullptr, sr.ct = hyper ? vct.d() : nullptr;
    sr.phase = 0;
    sr.t2 = t2_fused ? vct.d() : nullptr;           // (vct: free until phase 1 writes the r2-weighted twin into it)
    launch_strict_rowstats(sr, st);                 // p = A m, c = rowsum(T^2) - rowsum(A .* K^)   (:216, :218)
    sr.t2 = nullptr;
    if (!grads) return;
    rows_gemm(Ah.d(), Dm.d(), 1, 0, Pt.d());        // P~ = A (S Kuu^-1 - I)                      (:157-161)
    sr.phase = 1;
    launch_strict_rowstats(sr, st);                 // K^ a, rowsum(P~ .* K^) and their r2-weighted twins
  }

  // ------------------------------------------------------------------------------------------ row pass
  void row_pass() {
    const long long MM = (long long)M * M;
    const int ldz = Q * P;
    const bool want_hyper = (group_mask & (HMOGP_GROUP_HYPER | HMOGP_GROUP_Z)) != 0;
    const bool want_z = (group_mask & HMOGP_GROUP_Z) != 0;
    // (pools: see plan_pools())
    const long long ldn = ws_rows;
    if (!kuf_prefetched) HIP_TRY(hipMemsetAsync(stats.p, 0, sizeof(double) * nstats, st));   // (else: with the prefetch)
    const int tiles = (M + 127) / 128;
    const long long wtiles = (ws_rows + 127) / 128;
    const long long sK = ldn * M;                           // per-latent stride of the K^ / P~ workspaces
    const int ncb = (M + 127) / 128;
    for (auto& pl : pools) {
      const long long n = pl.back().off + pl.back().n;      // rows of this pool
      int* rw = use_windows ? winrow.as<int>() : nullptr;    // [Q][wtiles][2]
      int* cw = use_windows ? wincol.as<int>() : nullptr;    // [Q][ncb][2]
      const double* X = tasks[pl[0].t].X.d() + pl[0].r0 * P; // inputs of the pool's rows
      const bool prefetched = &pl == &pools[0] && kuf_prefetched;
      if (pl.size() > 1) {
        if (!prefetched) stage_pool_inputs(pl, st);
        X = Xws.d();
      }
      if (!prefetched && !small_rows) kuf_pool(pl, st);
      // Forward contraction for all latents (batched), row statistics fused into its epilogue; P~ itself is only stored
      // when the Z gradient (its one remaining consumer, colstats) is requested.  One launch per pool.
      const long long sPart = 4LL * FWD_PARTS * tiles * ldn;
      const long long clen = (long long)M * (2 + P);          // one column-statistics slab: [ r (M) | dZ (M*P) | s2 (M) ]
      // [r5] the r2-weighted statistic of the lengthscale gradient comes from the column statistics (E and x - z are in hand there),
      // not from two more row statistics of the forward epilogue; strict q(f) keeps its own (strict_rowstats_kernel, GPy's r2 form)
      static const bool col_sl_env = [] {   // HMOGP_COL_SL=0 (TIMING ONLY: sl is then missing from the lengthscale gradient)
        const char* e = getenv("HMOGP_COL_SL");
        return !(e && e[0] == '0');
      }();
      const bool col_sl = want_hyper && !strict && !small_rows && col_sl_env;
      // slabs of the column statistics: 256-row splits
      const long long csplit = col_split(n);
      const long long nsp = (n + csplit - 1) / csplit;        // slabs of the column statistics

      auto quad_segment = [&](const Seg& sg) {
        Task& k = tasks[sg.t];
        QuadArgs qa;
        qa.lik = k.lik, qa.lik_param = k.param, qa.dimf = k.dimf, qa.Q = Q, qa.N = sg.n;
        qa.y = k.Y.d() + sg.r0;
        qa.yaux = k.Yaux.p ? k.Yaux.d() + sg.r0 : nullptr;
        qa.p = vp.d() + sg.off, qa.c = vc.d() + sg.off;
        const bool row_sl = want_hyper && !col_sl;    // (small-model / strict paths: sl from the row statistics p~, c~)
        qa.pt = row_sl ? vpt.d() + sg.off : nullptr, qa.ct = row_sl ? vct.d() + sg.off : nullptr;
        qa.ldn = ldn;
        std::memset(qa.w, 0, sizeof(qa.w)), std::memset(qa.w0, 0, sizeof(qa.w0)), std::memset(qa.kap, 0, sizeof(qa.kap));
        std::memset(qa.var, 0, sizeof(qa.var));
        for (int q = 0; q < Q; ++q) {
          qa.var[q] = h_var[q];
          for (int j = 0; j < k.dimf; ++j) {
            qa.w[q][j] = h_W[q * Df + k.d0 + j];
            qa.w0[q][j] = h_W0[q * Df + k.d0 + j];
            qa.kap[q][j] = h_kap[q * Df + k.d0 + j];
          }
        }
        qa.scale = h_bs[sg.t];
        if (small_path) {     // (replayable from a captured graph: the mixing weights are read from the parameter block)
          qa.Wd = dW.d(), qa.W0d = dsmall.d() + oW0, qa.kapd = dkap.d(), qa.vard = dvar.d(), qa.scaled = dsmall.d() + oBs + sg.t;
          qa.Df = Df, qa.d0 = k.d0;
        }
        qa.quirks = quirks;
        if (strict && want_hyper) qa.pg = vpg.d() + sg.off, qa.cg = vcg.d() + sg.off;
        qa.alpha = valpha.d() + sg.off, qa.beta = vbeta.d() + sg.off;
        qa.alpha0 = valpha0.d() + sg.off, qa.beta0 = vbeta0.d() + sg.off;
        qa.partials = quadpart.d();
        launch_quad(qa, st);
        launch_reduce_rows(quadpart.d(), quad_blocks(k.lik, sg.n), k.nscal, k.offsets.as<long long>(), stats.d(), true, st);
      };
      // column statistics of rows [off, off + rows) on the second stream, after the quadrature of those rows (ev_fork)
      auto colstats_rows = [&](long long off, long long rows, long long slab_first) {
        HIP_TRY(hipEventRecord(ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(st2, ev_fork, 0));
        Scope sc(this, CAT_COLSTATS, 1, st2);
        ColBatch cb;
        cb.nq = Q, cb.sK = sK, cb.sA = M, cb.sV = ldn, cb.sZ = P, cb.sPart = nsp * clen, cb.sWin = 2 * ncb;
        // Blocks of the column statistics in flight beside the weighted Gram.  The Gram's 112 allocated registers per lane
        // leave room for one 64-register wave per SIMD, so these blocks run BESIDE two resident Gram blocks per CU and cost
        // them almost nothing -- as long as they do not saturate HBM: one block per row split (3125 x 6 at the headline size)
        // streams K^ and P~ at 4.5 TB/s for 8.7 ms, evicts the Gram's operand panels from the L2s and stretches it from
        // 39.3 to 45.4 ms; 192 blocks take 32 ms of the Gram's 40 at 1.2 TB/s and stretch it to 39.9 (profiles/
        // r03_colstats_cap.txt: step 126.8 -> 120.7 ms).  Bytes per Gram flop scale with 1 / M, so the cap does too.
        // [r5] the blocks also accumulate the r2-weighted statistic now, and the kernel is instantiated per (strict, statistic)
        // combination: with `want P~` a compile-time constant its row loop has no branch and a block streams 1.6x faster (192 blocks:
        // 26.2 ms instead of 42.7 at the headline size, the Gram unchanged at 39.7).  The cap is no longer proportional to 1 / M:
        // 256 blocks at M <= 512 (Gram 10.7 ms, column statistics 10.5: 33.5 ms per step instead of 34.9), 192 at M >= 1024
        // (profiles/r05_colstats_cap.txt).
        static const int cap_env = [] {   // HMOGP_COLSTATS_CAP=<blocks in flight> (0 = one block per row split)
          const char* e = getenv("HMOGP_COLSTATS_CAP");
          return e ? atoi(e) : -1;
        }();
        // (exact-zero windows: the banded Gram is short; the cap was sized for the dense one)
        // (P > 1: more arithmetic per byte -- a block streams 4.8 instead of 6.3 GB/s at P = 2 -- so proportionally more of them)
        // (strict q(f): the kernel streams a third matrix -- 256 blocks keep it as long as the Gram of A: 313.5 -> 310.7 ms at H)
        const int cap = cap_env >= 0 ? cap_env : (use_windows ? 0 : (int)(std::min(256.0, std::max(strict ? 256.0 : 192.0, 131072.0 / std::max(1, M))) * (1.0 + 0.35 * (P - 1))));
        launch_colstats(Kh.d() + off * M, Pt.d() + off * M, a.d(), valpha.d() + off, valpha0.d() + off, vbeta0.d() + off,
                        X + off * P, P, dZ.d(), ldz, rows, M, (int)csplit, want_z, colpart.d() + slab_first * clen, st2, cw, &cb, cap,
                        strict ? Ah.d() + off * M : nullptr, col_sl ? dell.d() : nullptr);
      };

      SmallRows sr;
      if (small_rows) {
        const long long nblk = (n + 63) / 64, slab_q = (long long)M * M + M + (long long)M * P;
        sr.M = M, sr.Q = Q, sr.P = P, sr.ldz = ldz, sr.hyper = want_hyper ? 1 : 0, sr.want_z = want_z ? 1 : 0, sr.n = n, sr.ldn = ldn;
        sr.X = X, sr.Z = dZ.d(), sr.var = dvar.d(), sr.ell = dell.d(), sr.C = C.d(), sr.a = a.d();
        sr.Kh = Kh.d(), sr.Pt = Pt.d(), sr.vp = vp.d(), sr.vc = vc.d(), sr.vpt = vpt.d(), sr.vct = vct.d();
        sr.alpha = valpha.d(), sr.beta = vbeta.d(), sr.alpha0 = valpha0.d(), sr.beta0 = vbeta0.d();
        smallslab.ensure(sizeof(double) * nblk * Q * slab_q);
        sr.slab = smallslab.d(), sr.stats = stats.d(), sr.NG = NG, sr.per_q = per_q, sr.oR = oR, sr.oDZ = oDZ;
        Scope sc(this, CAT_FWD, 1);
        launch_small_fwd(sr, st);      // K^ + P~ = K^ C_q + row statistics, one launch for all tasks and latents of the pool
      } else if (strict) {
        if (prefetched) HIP_TRY(hipStreamWaitEvent(st, ev_kuf, 0));
        strict_forward(n, X, want_hyper || want_z, want_hyper);
      } else
      {
        const long long off = 0, rows = n;
        if (prefetched) HIP_TRY(hipStreamWaitEvent(st, ev_kuf, 0));
        int nparts = 2;
        {
          // forward contraction for all latents (batched), row statistics fused into its epilogue; P~ itself is only
          // stored when the Z gradient (its one remaining consumer, colstats) is requested
          Scope sc(this, CAT_FWD, 1);
          GemmArgs g;
          g.A = Kh.d() + off * M, g.lda = M, g.a_kmajor = 0, g.sA = sK;
          // only the quadratic forms are wanted when neither the hyper-parameter nor the Z gradients are (SVI / VEM
          // E-steps): the triangular fold of C gives them with half the products
          const bool tri = !want_hyper && !want_z;
          g.B = tri ? Ctri.d() : C.d(), g.ldb = M, g.b_kmajor = 1, g.sB = MM, g.b_tri = tri ? 1 : 0;
          g.C = Pt.d() + off * M, g.ldc = M, g.sC = sK;
          g.M = (int)rows, g.N = M, g.K = M;
          g.nbatch = Q;
          g.role = 1;
          g.fs_part = fwdpart.d() + 4LL * FWD_PARTS * tiles * off, g.fs_sPart = sPart, g.fs_a = a.d(), g.fs_sA = M, g.fs_x = X + off * P;
          g.fs_z = dZ.d(), g.fs_sZ = P, g.fs_ldz = ldz, g.fs_P = P, g.fs_hyper = 0, g.fs_ell = dell.d();
          g.store_c = (want_z || want_hyper) ? 1 : 0;     // P~ is consumed by the column statistics (dZ and, [r5], sl)
          g.win = rw, g.win_stride = 2 * wtiles;
          nparts = launch_gemm_rowpass_or_general(g, st);
        }
        {
          Scope sc2(this, CAT_ROWSTATS, 1);  // sum of the per-column-tile partials of the fused row statistics
          launch_combine_parts(fwdpart.d() + 4LL * FWD_PARTS * tiles * off, nparts * tiles, rows, vp.d() + off, vc.d() + off,
                               nullptr, nullptr, st, Q, sPart, ldn);
        }
      }
      // small models: every segment of the pool in ONE quadrature launch, its block partials summed by small_red_kernel
      SmallQuadRed qred;
      long long qblocks = 0;
      for (auto& sg : pl) qblocks += quad_blocks(tasks[sg.t].lik, sg.n);
      // [r5] ... and on the regular path too where the pool is SHORT (minibatches, rank shares: four launches of a few dozen blocks
      // + four reductions were 0.17 ms between the forward and the Gram of an 8192-row step) and its likelihood set has an
      // instantiation of its own; the full-batch sizes keep one launch per task (each with its own register allocation)
      static const bool qm_regular_env = [] {   // HMOGP_QUAD_MULTI_REGULAR=0: one quadrature launch per task on the regular path
        const char* e = getenv("HMOGP_QUAD_MULTI_REGULAR");
        return !(e && e[0] == '0');
      }();
      bool quad_multi = small_rows && (int)pl.size() <= HMOGP_QUAD_MULTI && qblocks <= 2048;
      const bool qm_regular = !small_rows && !strict && qm_regular_env && pl.size() >= 2 && (int)pl.size() <= HMOGP_QUAD_MULTI &&
                              qblocks <= 2048;
      if (quad_multi || qm_regular) {
        QuadMulti qm;
        qm.nseg = (int)pl.size(), qm.Q = Q, qm.Df = Df, qm.ldn = ldn;
        const bool row_sl = want_hyper && !col_sl;    // (sl from the row statistics p~, c~: small-model path only)
        qm.p = vp.d(), qm.c = vc.d(), qm.pt = row_sl ? vpt.d() : nullptr, qm.ct = row_sl ? vct.d() : nullptr;
        qm.Wd = dW.d(), qm.W0d = dsmall.d() + oW0, qm.kapd = dkap.d(), qm.vard = dvar.d(), qm.scale_base = dsmall.d() + oBs;
        qm.quirks = quirks;
        qm.alpha = valpha.d(), qm.beta = vbeta.d(), qm.alpha0 = valpha0.d(), qm.beta0 = vbeta0.d(), qm.partials = quadpart.d();
        long long part = 0;
        for (size_t i = 0; i < pl.size(); ++i) {
          const Seg& sg = pl[i];
          Task& k = tasks[sg.t];
          QuadSeg& g = qm.seg[i];
          g.lik = k.lik, g.dimf = k.dimf, g.d0 = k.d0, g.t = sg.t, g.lik_param = k.param, g.N = sg.n, g.off = sg.off;
          g.y = k.Y.d() + sg.r0, g.yaux = k.Yaux.p ? k.Yaux.d() + sg.r0 : nullptr;
          auto& r = qred.s[qred.nseg++];
          r.part = quadpart.d() + part, r.nrows = quad_blocks(k.lik, sg.n), r.nscal = k.nscal, r.off = k.offsets.as<long long>();
          part += r.nrows * k.nscal;
        }
        if (quad_multi || quad_multi_specialised(qm)) {
          Scope sc(this, CAT_QUAD, quad_multi ? 1 : 2);
          launch_quad_multi(qm, st);
          if (!quad_multi) launch_reduce_rows_multi(qred, stats.d(), st);   // (small models: summed by small_red_kernel)
        } else {
          Scope sc(this, CAT_QUAD, 2 * (int)pl.size());
          for (auto& sg : pl) quad_segment(sg);
        }
      } else {
        Scope sc(this, CAT_QUAD, 2 * (int)pl.size());
        for (auto& sg : pl) quad_segment(sg);
      }
      // The column statistics (HBM-bound: K^ and P~ streamed once) run on the second stream BESIDE the weighted Gram: both
      // only need the row weights of the quadrature and write disjoint parts of the bundle.  (Measured alternative: the
      // column statistics of segment i beside the forward contraction of segment i + 1 -- the Gram gains 4.0 ms, the
      // forward contractions lose 5.7 ms: an HBM-saturating kernel costs an FP64-MFMA GEMM beside it about its own
      // stand-alone time either way.)
      if (small_rows) {
        Scope sc(this, CAT_GRAM, 2);
        launch_small_bwd(sr, st, quad_multi ? &qred : nullptr);   // H_q, r_q, dZ_q: block partials + their ordered sum into the bundle
        continue;
      }
      colstats_rows(0, n, 0);
      {
        // H_q += K^T diag(beta) K^ for all latents (svmogp_inf.py:145-147 summed over d)
        const int ksplit = use_windows ? std::min(8, gram_ksplit(n, M)) : gram_ksplit(n, M);
        slabs.ensure(sizeof(double) * MM * ksplit * Q, true);
        GemmArgs g;
        // (strict q(f): the Gram of A = K^ Kuu^-1 IS dVE_dS, svmogp_inf.py:145-148)
        g.A = strict ? Ah.d() : Kh.d(), g.lda = M, g.a_kmajor = 1, g.sA = sK;
        g.B = g.A, g.ldb = M, g.b_kmajor = 1, g.sB = sK;
        g.kscale = vbeta.d(), g.sS = ldn;
        g.C = slabs.d(), g.ldc = M, g.sC = MM * ksplit;
        g.M = g.N = M, g.K = (int)n;
        g.nbatch = Q;
        g.lower_only = 1;
        g.ksplit = ksplit, g.sSplit = MM;
        g.role = 2;
        static const int bal = [] {   // HMOGP_DIAG_BALANCE=0: static sub-tile assignment on the diagonal tiles (A/B runs)
          const char* e = getenv("HMOGP_DIAG_BALANCE");
          return e ? atoi(e) : 1;
        }();
        g.diag_balance = bal;
        g.win = cw, g.win_stride = 2 * ncb;
        {
          Scope sc(this, CAT_GRAM, 1);
          launch_gemm_rowpass_or_general(g, st);
        }
        {
          Scope sc(this, CAT_COLSTATS, 1, st2);   // all 256-row slabs of the pool -> bundle
          launch_reduce_slabs(colpart.d(), (int)nsp, clen, (long long)M * (1 + P), Hq(0) + oR, true, st2, Q, nsp * clen, per_q);
          if (col_sl) {   // per-column s2 -> [Q][M] -> added into sl_q in a fixed order
            launch_reduce_slabs(colpart.d() + (long long)M * (1 + P), (int)nsp, clen, M, colred.d(), false, st2, Q, nsp * clen, M);
            launch_sum_cols(colred.d(), Q, M, Hq(0) + oSL, per_q, st2);
          }
        }
        HIP_TRY(hipEventRecord(ev_col, st2));
        Scope sc2(this, CAT_COLSTATS, 1);  // row-range slabs -> bundle (accounted with the column statistics)
        launch_reduce_slabs_lower(slabs.d(), ksplit, M, Hq(0), true, st, Q, MM * ksplit, per_q);
      }
      HIP_TRY(hipStreamWaitEvent(st, ev_col, 0));      // the workspaces are reused by the next pool
    }
    // (H_q holds its lower triangle only from here to hmogp_step_finish, which mirrors it: the exchange step of a
    // multi-GPU run all-reduces the triangle, wire_pack / wire_unpack)
  }

  // bundle <-> wire format, synchronous at return (the caller's all-reduce runs on another stream / library)
  void wire_copy(int dir) {
    if (!began) throw EngineError{HMOGP_E_STATE, "wire pack / unpack outside hmogp_step_begin .. hmogp_step_finish"};
    HIP_TRY(hipSetDevice(device));
    wire.ensure(sizeof(double) * nwire, true);
    launch_wire_copy(stats.d(), wire.d(), NG, Q, M, per_q, dir, st);
    HIP_TRY(hipStreamSynchronize(st));
  }

  void decide_mode(const hmogp_params* p) {
      static const int small_env = [] {   // HMOGP_SMALL_MODE=0|1: force the small-problem mode off / on (A/B runs)
        const char* e = getenv("HMOGP_SMALL_MODE");
        return e ? atoi(e) : -1;
      }();
      long long rows_eval = 0;
      for (int t = 0; t < T && p; ++t) {
        const long long b = p->row_begin ? p->row_begin[t] : 0, e = p->row_end ? p->row_end[t] : tasks[t].N;
        rows_eval += std::max<long long>(0, e - b);
      }
      // (strict q(f) -- config flag or this evaluation's HMOGP_EVAL_STRICT_QF -- runs on the regular kernels: the fused small-model
      //  kernels carry the explicit-inverse algebra only)
      strict = strict_cfg || (p && (p->eval_flags & HMOGP_EVAL_STRICT_QF) != 0);
      skip_g_L = p && (p->eval_flags & HMOGP_EVAL_NO_G_L) != 0;
      if (strict && use_windows) throw EngineError{HMOGP_E_INVALID, "strict q(f) and HMOGP_CFG_EXACT_ZERO_WINDOWS exclude each other"};
      const bool want_small = !no_small && !strict && (small_env >= 0 ? small_env != 0 : (M <= 128 && rows_eval <= 65536 && !use_windows && !st2_masked));
      if (want_small != small_mode) {     // (rare: drain the queues the previous evaluations used before re-wiring them)
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipStreamSynchronize(st2_own));
        HIP_TRY(hipStreamSynchronize(st3_own));
        small_mode = want_small;
        st2 = small_mode ? st : st2_own;
        st3 = small_mode ? st : st3_own;
      }
      static const int path_env = [] {   // HMOGP_SMALL_PATH=0: keep the regular kernels in small-problem mode (A/B runs)
        const char* e = getenv("HMOGP_SMALL_PATH");
        return e ? atoi(e) : 1;
      }();
      // (a SHARDED step -- hmogp_elbo_grad_sharded, or the split form hmogp_step_begin ... hmogp_step_finish whose bundle the caller
      //  exchanges with whatever it has: the library's communicator, torch.distributed, MPI -- must not choose its path from this
      //  rank's row count: every rank takes the regular kernels, so that the replicated M x M algebra, and with it the never
      //  re-synchronised resident q(u) replicas, round alike on all ranks.  ADVICE r4.  Plain hmogp_elbo_grad is unaffected.)
      small_path = small_mode && M <= HMOGP_SMALL_M && path_env != 0 && !small_veto && !sharded_call;
      small_info_pending = false;
      static const int rows_env = [] {   // HMOGP_SMALL_ROWS=0: the regular row-pass kernels behind the fused M x M kernels (A/B runs)
        const char* e = getenv("HMOGP_SMALL_ROWS");
        return e ? atoi(e) : 1;
      }();
      small_rows = small_path && rows_env != 0;
    }

  bool sharded_call = false;
  bool skip_g_L = false;        // HMOGP_EVAL_NO_G_L of this evaluation
  void begin(const hmogp_params* p, bool sync = true, bool will_exchange = false) {
    HIP_TRY(hipSetDevice(device));
    sharded_call = will_exchange || sync;      // (hmogp_step_begin is the first half of a split, i.e. exchanged, step)
    began = false, exchanged = false;
    spans.clear();  // a failed evaluation may have left unmatched timing spans behind
    pool_used = 0;
    for (int c = 0; c < NCAT; ++c) ms[c] = 0.0, launches[c] = 0;
    decide_mode(p);
    info_early = sync || will_exchange;
    upload_params(p);
    HIP_TRY(hipEventRecord(ev_begin0, st));
    plan_pools();
    ensure_strict_workspace();
    u_algebra();
    row_pass();
    HIP_TRY(hipEventRecord(ev_begin1, st));
    // hmogp_step_begin returns with the bundle complete (the caller all-reduces it); the fused hmogp_elbo_grad goes straight
    // on to enqueue the post-processing behind the row pass -- no host round trip, no launch latency in the tail
    if (sync) HIP_TRY(hipStreamSynchronize(st));
    if (small_path && (sync || will_exchange)) {   // callers that exchange the bundle must know NOW whether the factorisation held
      if (!sync) HIP_TRY(hipStreamSynchronize(st));
      if (small_failed()) {
        small_veto = true;
        try {
          begin(p, sync, will_exchange);
        } catch (...) {
          small_veto = false;
          throw;
        }
        small_veto = false;
        return;
      }
    }
    began = true;
  }
  // after a synchronisation behind u_small_kernel: did a latent's plain factorisation fail?  (forced rung: an error)
  bool small_failed() {
    if (!small_info_pending) return false;
    small_info_pending = false;
    bool failed = false;
    for (int q = 0; q < Q; ++q)
      if ((info_early ? h_info[q] : (int)hstage[fl.n_stage + q]) != 0) {
        if (rung_request[q] != -2) throw EngineError{HMOGP_E_NOT_PD, "Cholesky failed at the forced jitter rung"};
        failed = true;
      }
    return failed;
  }

  // ------------------------------------------------------------------------------------------ finish
  // what one evaluation returns through the single D2H staging block: the small results (head of the bundle, KL partials,
  // per-latent tails, K_uu-side rows) gathered device-side; on the small-model path the q(u) gradients ride in the same block
  struct FinLayout {
    bool want_qu = false, want_hz = false, qu_out = false;
    size_t n_hg = 0, n_kl = 0, n_tail = 0, n_row = 0, n_all = 0, n_gmu = 0, n_gl = 0, n_stage = 0;
  } fl;
  void fin_layout(const hmogp_outputs* out) {
    const long long Mtri = (long long)M * (M + 1) / 2;
    fl.want_qu = (group_mask & HMOGP_GROUP_QU) != 0 || out->dL_dS != nullptr;
    fl.want_hz = (group_mask & (HMOGP_GROUP_HYPER | HMOGP_GROUP_Z)) != 0;
    fl.n_hg = NG, fl.n_kl = (size_t)Q * KL_BLOCKS * 6, fl.n_tail = (size_t)Q * (per_q - oDZ);
    fl.n_row = fl.want_hz ? (size_t)Q * M * (2 + P) : 0, fl.n_all = fl.n_hg + fl.n_kl + fl.n_tail + fl.n_row;
    fl.qu_out = small_path && fl.want_qu && (group_mask & HMOGP_GROUP_QU) != 0;
    fl.n_gmu = fl.qu_out ? (size_t)M * Q : 0, fl.n_gl = fl.qu_out ? (size_t)Mtri * Q : 0;
    fl.n_stage = fl.n_all + fl.n_gmu + fl.n_gl;
    dstage.ensure(sizeof(double) * fl.n_stage);
    if (hstage_cap < fl.n_stage + HMOGP_MAXQ) {      // (+ the info words of the small path)
      if (hstage) (void)hipHostFree(hstage);
      hstage = nullptr, hstage_cap = 0, hstage_dev = nullptr;
      drop_graphs(true);                             // (captured kernels hold the old block's address)
      HIP_TRY(hipHostMalloc((void**)&hstage, sizeof(double) * (fl.n_stage + HMOGP_MAXQ), hipHostMallocDefault));
      HIP_TRY(hipHostGetDevicePointer((void**)&hstage_dev, hstage, 0));
      hstage_cap = fl.n_stage + HMOGP_MAXQ;
    }
  }
  void finish(hmogp_outputs* out) {
    finish_enqueue(out);
    finish_tail(out);
  }
  void finish_enqueue(hmogp_outputs* out) {
    if (!began) throw EngineError{HMOGP_E_STATE, "hmogp_step_finish without hmogp_step_begin"};
    if (!out) throw EngineError{HMOGP_E_INVALID, "null outputs"};
    HIP_TRY(hipSetDevice(device));
    const long long MM = (long long)M * M, Mtri = (long long)M * (M + 1) / 2;
    fin_layout(out);
    const bool want_qu = fl.want_qu, want_hz = fl.want_hz, qu_out = fl.qu_out;
    const size_t n_hg = fl.n_hg, n_kl = fl.n_kl, n_row = fl.n_row, n_all = fl.n_all, n_gmu = fl.n_gmu, n_stage = fl.n_stage;
    HIP_TRY(hipEventRecord(ev_fin0, st));
    HIP_TRY(hipStreamWaitEvent(st, ev_join, 0));   // the S^-1 chain of hmogp_step_begin (third stream) used HK / G as scratch
    if (small_path) {
      // M <= 64: the whole post-processing of the bundle in ONE kernel (one block per latent, matrices in LDS), then the K_uu-side
      // row sums; q(u) gradients leave on the same (only) stream
      Scope sc(this, CAT_MM, 2);
      SmallF f;
      f.M = M, f.Q = Q, f.want_qu = want_qu ? 1 : 0, f.want_hz = want_hz ? 1 : 0, f.per_q = per_q, f.oR = oR;
      f.H = Hq(0), f.Hfull = Hq(0), f.Kuui = Kuui.d(), f.KiS = KiS.d(), f.KSK = KSK.d(), f.Sqi = Sqi.d(), f.L = L.d(), f.a = a.d();
      f.G = G.d(), f.GSK = GSK.d(), f.dLdS = dLdS.d(), f.dKmm = dKmm.d(), f.Kr = Kr.d(), f.gL = gL.d(), f.gmu = gmu.d();
      if (qu_out) f.gmu2 = dstage.d() + n_all, f.gL2 = dstage.d() + n_all + n_gmu;
      if (want_hz) f.Z = dZ.d(), f.var = dvar.d(), f.ell = dell.d(), f.P = P, f.ldz = Q * P, f.rowout = rowout.d();
      // the last block to finish writes every small result straight into the page-locked host block
      f.stage = hstage_dev, f.g_stats = stats.d(), f.g_kl = klout.d(), f.g_extra = dstage.d() + n_all, f.g_info = dinfo.as<int>();
      f.n_hg = (long long)n_hg, f.n_kl = (long long)n_kl, f.n_tail = per_q - oDZ, f.oDZ = oDZ, f.n_row = (long long)n_row;
      f.n_extra = (long long)(n_stage - n_all), f.NG = NG, f.counter = dinfo.as<int>() + 2 * HMOGP_MAXQ;
      launch_finish_small(f, st);    // (+ the K_zz-weighted row sums of dL_dKmm: kzz_rows_kernel's arithmetic)
      HIP_TRY(hipEventRecord(ev_join, st));
    } else
    {
      Scope sc(this, CAT_MM, 0);
      launch_mirror_lower(Hq(0), Q, M, per_q, st);                   // the row pass / the exchange fill the lower triangle
      if (strict) {   // the bundle already holds dVE_dS = A^T diag(beta) A and dVE_dmu = A^T alpha (svmogp_inf.py:144-148)
        HIP_TRY(hipMemcpy2DAsync(G.p, sizeof(double) * MM, Hq(0), sizeof(double) * per_q, sizeof(double) * MM, Q, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpy2DAsync(Kr.p, sizeof(double) * M, Hq(0) + oR, sizeof(double) * per_q, sizeof(double) * M, Q, hipMemcpyDeviceToDevice, st));
      } else {
        mm(Hq(0), false, Kuui.d(), true, HK.d(), 1.0, per_q);          // H K^-1
        mm(Kuui.d(), false, HK.d(), true, G.d(), 1.0, -1, -1, nullptr, 0, 0, true);  // G = K^-1 H K^-1 (dVE_dS, svmogp_inf.py:148):
        launch_mirror_lower(G.d(), Q, M, MM, st);                      // symmetric -> lower tiles only, then mirrored
        launch_gemv_batched(Kuui.d(), Hq(0) + oR, Kr.d(), Q, M, per_q, 1, st);  // K^-1 r  (dVE_dmu, :144)
      }
      // two independent tails: the K_uu-side gradients stay on the main stream, the q(u) gradients and the KL terms
      // go to the second one
      HIP_TRY(hipEventRecord(ev_fork, st));
      HIP_TRY(hipStreamWaitEvent(st3, ev_fork, 0));
      if (want_qu) {
        launch_dlds(G.d(), Kuui.d(), Sqi.d(), dLdS.d(), MM * Q, st3);
        HIP_TRY(hipEventRecord(ev_S, st3));
        if (!skip_g_L) {   // (HMOGP_EVAL_NO_G_L: a natural-gradient E-step consumes dL/dS and dL/dm only)
          mm(dLdS.d(), false, L.d(), true, tmpA.d(), 1.0, -1, -1, st3, 0, +1);  // dL_dS L (:175-177), L lower
          launch_pack_gl(tmpA.d(), gL.d(), Q, M, st3);
        }
        launch_gmu(Kr.d(), a.d(), gmu.d(), Q, M, st3);
      }
      if (want_hz) {
        // G S K^-1 (tmp_dv, :151), released together with dL/dS L of the q(u) tail: the two products share the matrix cores.
        // The 12.6 MB D2H copy of that tail waits for both: a product that is still running when the copy starts does
        // not finish before the copy does (363-438 us instead of 121 measured, whichever stream or priority it is on);
        // the small kernels behind it run beside the copy.
        if (want_qu) HIP_TRY(hipStreamWaitEvent(st, ev_S, 0));
        // [r5] formed TRANSPOSED, K^-1 S G = (G S K^-1)^T (G is exactly symmetric; dL_dKmm only ever uses GSK + GSK^T): the
        // operand layouts of this form take the k-major-B kernel variant, 120 instead of 212 us at M = 1024, Q = 3
        mm(KiS.d(), false, G.d(), true, GSK.d());
        HIP_TRY(hipEventRecord(ev_gsk, st));
        if (want_qu) HIP_TRY(hipStreamWaitEvent(st3, ev_gsk, 0));
      }
      if (want_qu) {
        // the large gradient leaves on this stream as soon as it exists, beside the K_uu-side tail of the main stream
        if (out->g_L_u && (group_mask & HMOGP_GROUP_QU) && !skip_g_L)
          HIP_TRY(hipMemcpyAsync(out->g_L_u, gL.p, sizeof(double) * Mtri * Q, hipMemcpyDeviceToHost, st3));
        if (out->g_m_u && (group_mask & HMOGP_GROUP_QU))
          HIP_TRY(hipMemcpyAsync(out->g_m_u, gmu.p, sizeof(double) * M * Q, hipMemcpyDeviceToHost, st3));
      }
      HIP_TRY(hipEventRecord(ev_join, st3));
      if (want_hz) {
        launch_dkmm(G.d(), GSK.d(), Kuui.d(), KSK.d(), Kr.d(), a.d(), dKmm.d(), Q, M, st);
        launch_kzz_rows(dKmm.d(), dZ.d(), Q * P, P, dvar.d(), dell.d(), Q, M, rowout.d(), st);
      }
      HIP_TRY(hipStreamWaitEvent(st, ev_join, 0));
    }
    // ---- device -> host ------------------------------------------------------------------------------
    // the small results (head of the bundle, KL partials, per-latent tails, K_uu-side rows) are gathered device-side and
    // leave in ONE copy into a page-locked buffer: ten separate pageable copies cost 0.3 ms of gaps
    if (!small_path) {
      // [r5] gathered STRAIGHT into the page-locked host block (its device-side address), like the small-model path: the separate
      // D2H copy command behind the gather kernel started 240-390 us after it (rocprofv3 timelines of H and C3: the copy waited
      // for the 12.6 MB g_L_u transfer of the other stream to drain) -- 5 % of a minibatch step for 10 KB of results.
      static const bool direct = [] {   // HMOGP_GATHER_DIRECT=0: gather into HBM + hipMemcpyAsync as before (A/B runs)
        const char* e = getenv("HMOGP_GATHER_DIRECT");
        return !(e && e[0] == '0');
      }();
      double* d = direct ? hstage_dev : dstage.d();
      launch_gather_small(stats.d(), (long long)n_hg, klout.d(), (long long)n_kl, per_q, oDZ, per_q - oDZ, Q, rowout.d(),
                          (long long)n_row, d, st);
      if (!direct) HIP_TRY(hipMemcpyAsync(hstage, d, sizeof(double) * n_stage, hipMemcpyDeviceToHost, st));
    }
    if (out->dL_dS) HIP_TRY(hipMemcpyAsync(out->dL_dS, dLdS.p, sizeof(double) * MM * Q, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(ev_fin1, st));
    (void)n_gmu, (void)n_all;
  }
  // everything behind the last enqueued operation of an evaluation: the one host synchronisation, then the host assembly
  void finish_tail(hmogp_outputs* out) {
    const long long Mtri = (long long)M * (M + 1) / 2;
    const bool want_qu = fl.want_qu, want_hz = fl.want_hz, qu_out = fl.qu_out;
    const size_t n_hg = fl.n_hg, n_kl = fl.n_kl, n_tail = fl.n_tail, n_all = fl.n_all, n_gmu = fl.n_gmu, n_gl = fl.n_gl;
    const double *hg = hstage, *hkl = hstage + n_hg, *htail = hstage + n_hg + n_kl, *hrow = hstage + n_hg + n_kl + n_tail;
    const bool qu = (group_mask & HMOGP_GROUP_QU) != 0;
    if (out->g_m_u && !qu) std::memset(out->g_m_u, 0, sizeof(double) * M * Q);      // (copied on the second stream otherwise)
    if (out->g_L_u && (!qu || (skip_g_L && !small_path))) std::memset(out->g_L_u, 0, sizeof(double) * Mtri * Q);
    if (exchanged && comm) wait_exchanged();          // a collective is in flight: watchdog instead of a blind wait
    HIP_TRY(hipStreamSynchronize(st));
    if (small_path && small_failed()) {               // a latent needs GPy's jitter ladder: the regular path owns it
      spans.clear(), pool_used = 0;
      began = false;
      throw RetryRegular{};
    }
    collect_spans();
    if (qu_out) {      // small-model path: the q(u) gradients arrived in the staging block
      if (out->g_m_u) std::memcpy(out->g_m_u, hstage + n_all, sizeof(double) * n_gmu);
      if (out->g_L_u) std::memcpy(out->g_L_u, hstage + n_all + n_gmu, sizeof(double) * n_gl);
    }
    float f0 = 0.f, f1 = 0.f;
    if (!via_graph) {     // (events recorded by graph nodes are not read back: a replayed evaluation reports no device time)
      (void)hipEventElapsedTime(&f0, ev_begin0, ev_begin1);
      (void)hipEventElapsedTime(&f1, ev_fin0, ev_fin1);
    }
    ms[CAT_TOTAL] = f0 + f1 + ms[CAT_EXCHANGE];

    // ---- host assembly (svmogp.py:101-166) -----------------------------------------------------------
    double KL = 0.0, ninf = 0.0;
    for (int q = 0; q < Q; ++q) {
      double k[5] = {0, 0, 0, 0, 0};
      for (int b = 0; b < KL_BLOCKS; ++b)
        for (int i = 0; i < 5; ++i) k[i] += hkl[((size_t)q * KL_BLOCKS + b) * 5 + i];
      const double klq = 0.5 * k[0] + 0.5 * k[1] - 0.5 * M + k[2] - k[3];  // svmogp_inf.py:245-249
      KL += klq;
      if (out->kl) out->kl[q] = klq;
      ninf += k[4];
    }
    if (out->elbo) out->elbo[0] = hg[0] - KL;
    // [r5] condition estimate variance * max_i (K_uu^-1)_ii (a lower bound of cond(K_uu + jitter), 30-150x below it on RBF matrices)
    // and the flag that says which mode can still be trusted with it: the explicit-C_q path keeps element-wise 1e-5 to cond ~ 1e4
    // (estimate ~ 5e2), the strict path through everything GPy's jitter rung 0 leaves behind (cond ~ 1e7 ... 2e7, estimate 4e5 ... 8e5:
    // threshold 1e6) -- tools/ladder_sweep.py, DESIGN 6a
    bool ill = false;
    for (int q = 0; q < Q; ++q) {
      double kmax = 0.0;
      for (int b = 0; b < KL_BLOCKS; ++b) kmax = std::max(kmax, hkl[(size_t)Q * KL_BLOCKS * 5 + (size_t)q * KL_BLOCKS + b]);
      const double est = kmax * h_var[q];
      if (out->cond_est) out->cond_est[q] = est;
      ill = ill || est > (strict ? 1e6 : 5e2);
    }
    if (out->flags) out->flags[0] = ((hg[1] > 0.0) ? HMOGP_FLAG_V_NEGATIVE : 0u) | (ill ? HMOGP_FLAG_ILL_CONDITIONED : 0u);
    if (out->rung) std::copy(rung.begin(), rung.end(), out->rung);
    const bool hy = (group_mask & HMOGP_GROUP_HYPER) != 0, zz = (group_mask & HMOGP_GROUP_Z) != 0;
    const double* sgv = &hg[2];
    for (int q = 0; q < Q; ++q) {
      const double* tail = &htail[q * (per_q - oDZ)];
      const double* dZs = tail;
      const double sa = tail[oSA - oDZ], sl = tail[oSL - oDZ];
      const double* swk = tail + (oSWK - oDZ);
      double s1 = 0.0, s2 = 0.0;
      if (want_hz)
        for (int m = 0; m < M; ++m) {
          s1 += hrow[((size_t)q * M + m) * (2 + P)];
          s2 += hrow[((size_t)q * M + m) * (2 + P) + 1];
        }
      const double var = h_var[q], ell = h_ell[q];
      if (out->g_variance) {
        double g = 0.0;
        if (hy) {
          g = s1 / var + sa / var;
          for (int d = 0; d < Df; ++d) g += (h_W0[q * Df + d] * h_W0[q * Df + d] + h_kap0[q * Df + d]) * sgv[d];
        }
        out->g_variance[q] = g;
      }
      if (out->g_lengthscale) out->g_lengthscale[q] = hy ? (s2 / ell + sl / ell) : 0.0;
      for (int d = 0; d < Df; ++d) {
        // util.py:230 + :252 (quirk Q4: the K_ff-diagonal part is W sum(gv); the true value is 2 W variance sum(gv))
        const double wdiag = (quirks & HMOGP_QUIRK_W_DIAG) ? h_W[q * Df + d] * sgv[d] : 2.0 * h_W[q * Df + d] * var * sgv[d];
        if (out->g_W) out->g_W[q * Df + d] = hy ? (wdiag + swk[d]) : 0.0;
        // util.py:231 (quirk Q5: sum(gv); the true value is variance sum(gv))
        if (out->g_kappa) out->g_kappa[q * Df + d] = hy ? ((quirks & HMOGP_QUIRK_KAPPA_DIAG) ? sgv[d] : var * sgv[d]) : 0.0;
      }
      if (out->g_Z)
        for (int m = 0; m < M; ++m)
          for (int p = 0; p < P; ++p)
            out->g_Z[(size_t)m * Q * P + q * P + p] =
                zz ? (dZs[m * P + p] / (ell * ell) + hrow[((size_t)q * M + m) * (2 + P) + 2 + p] / (ell * ell)) : 0.0;
    }
    evaluated = true;
    have_qu_grads = want_qu;
    began = false;
    if (ninf > 0.0) throw EngineError{HMOGP_E_SQI_UNSTABLE, "Sqi: Cholesky representation unstable"};
  }

  // ------------------------------------------------------------------------------------------ consumers
  void posterior_u(double* wv, double* winv) {
    if (!evaluated && !began) throw EngineError{HMOGP_E_STATE, "no evaluation to take the posterior from"};
    HIP_TRY(hipSetDevice(device));
    const long long MM = (long long)M * M;
    if (wv) HIP_TRY(hipMemcpyAsync(wv, a.p, sizeof(double) * Q * M, hipMemcpyDeviceToHost, st));
    if (winv) {
      launch_sub(Kuui.d(), KSK.d(), tmpA.d(), MM * Q, st);  // K^-1 - K^-1 S K^-1 (GPy Posterior.woodbury_inv)
      HIP_TRY(hipMemcpyAsync(winv, tmpA.p, sizeof(double) * MM * Q, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
  }

  // ---- device-resident q(u) + Adadelta (SVI loop; util.py:321-329, svmogp.py:188-199) ------------
  void qu_load(const double* m_u, const double* L_flat) {
    if (!m_u || !L_flat) throw EngineError{HMOGP_E_INVALID, "null q(u) arrays"};
    HIP_TRY(hipSetDevice(device));
    const size_t nm = sizeof(double) * M * Q, nl = sizeof(double) * ((long long)M * (M + 1) / 2) * Q;
    HIP_TRY(hipMemcpyAsync(dmu.p, m_u, nm, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dLflat.p, L_flat, nl, hipMemcpyHostToDevice, st));
    for (DevBuf* b : {&ad_gms_m, &ad_sms_m, &ad_step_m, &ad_pend_m}) {
      b->ensure(nm);
      HIP_TRY(hipMemsetAsync(b->p, 0, nm, st));
    }
    for (DevBuf* b : {&ad_gms_L, &ad_sms_L, &ad_step_L, &ad_pend_L}) {
      b->ensure(nl);
      HIP_TRY(hipMemsetAsync(b->p, 0, nl, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    qu_resident = true;
  }
  void qu_read(double* m_u, double* L_flat) {
    if (!qu_resident) throw EngineError{HMOGP_E_STATE, "no resident q(u)"};
    HIP_TRY(hipSetDevice(device));
    if (m_u) HIP_TRY(hipMemcpyAsync(m_u, dmu.p, sizeof(double) * M * Q, hipMemcpyDeviceToHost, st));
    if (L_flat) HIP_TRY(hipMemcpyAsync(L_flat, dLflat.p, sizeof(double) * ((long long)M * (M + 1) / 2) * Q, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // phase 0: momentum move before the gradient evaluation; phase 1: update from the gradients the last evaluation left in
  // gmu / gL (objective = -ELBO: sign -1), or from a zero gradient when that evaluation did not include the q(u) group
  void qu_adadelta(int phase, double rate, double m, double d, double omd, double o) {
    if (!qu_resident) throw EngineError{HMOGP_E_STATE, "no resident q(u)"};
    if (phase == 1 && !evaluated) throw EngineError{HMOGP_E_STATE, "Adadelta update without a finished evaluation"};
    if (phase == 1 && skip_g_L && !small_path && (group_mask & HMOGP_GROUP_QU) != 0)
      throw EngineError{HMOGP_E_STATE, "the last evaluation ran with HMOGP_EVAL_NO_G_L: it left no gradient of q(u)'s factor"};
    HIP_TRY(hipSetDevice(device));
    const long long nm = (long long)M * Q, nl = ((long long)M * (M + 1) / 2) * Q;
    const bool has = phase == 1 && (group_mask & HMOGP_GROUP_QU) != 0;
    launch_adadelta(dmu.d(), ad_gms_m.d(), ad_sms_m.d(), ad_step_m.d(), ad_pend_m.d(), has ? gmu.d() : nullptr, -1.0, nm, phase, rate, m, d, omd, o, st);
    launch_adadelta(dLflat.d(), ad_gms_L.d(), ad_sms_L.d(), ad_step_L.d(), ad_pend_L.d(), has ? gL.d() : nullptr, -1.0, nl, phase, rate, m, d, omd, o, st);
    // (no host synchronisation: every consumer of the resident q(u) is ordered behind this stream -- the next evaluation's q(u)
    //  chain on the third stream waits for ev_qu, hmogp_qu_read / hmogp_qu_natgrad run on this stream)
    HIP_TRY(hipEventRecord(ev_qu, st));
  }

  // Inner-protocol debug export (include/hetmogp_hip.h: hmogp_debug_raw_grads): the gradient dictionary of
  // SVMOGPInf.inference (svmogp_inf.py:107) rebuilt from what the last evaluation left in HBM -- dKmm, a, P~ of the one
  // pool, p / c row statistics -- plus one more quadrature pass that writes the per-function d ve/dm, d ve/dv rows.
  void debug_raw(double* o_kmm, double* o_kmn, double* o_kdiag) {
    if (!evaluated) throw EngineError{HMOGP_E_STATE, "no finished evaluation"};
    if ((group_mask & HMOGP_GROUP_ALL) != HMOGP_GROUP_ALL) throw EngineError{HMOGP_E_STATE, "debug export needs group_mask = HMOGP_GROUP_ALL"};
    if (pools.size() != 1) throw EngineError{HMOGP_E_STATE, "debug export needs all rows in one pool (small N)"};
    HIP_TRY(hipSetDevice(device));
    const long long MM = (long long)M * M, ldn = ws_rows;
    if (o_kmm) HIP_TRY(hipMemcpyAsync(o_kmm, dKmm.p, sizeof(double) * MM * Q, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!o_kmn && !o_kdiag) return;
    const auto& pl = pools[0];
    std::vector<long long> nt(T, 0), off(T, 0);
    for (auto& sg : pl) {
      if (nt[sg.t] == 0) off[sg.t] = sg.off;
      nt[sg.t] += sg.n;   // a task's segments are contiguous inside the pool
    }
    long long nmax = 1;
    for (int t = 0; t < T; ++t) nmax = std::max(nmax, nt[t]);
    DevBuf gm, gv, tile;
    std::vector<DevBuf> gmt(T), gvt(T);
    for (auto& sg : pl) {
      Task& k = tasks[sg.t];
      gmt[sg.t].ensure(sizeof(double) * nt[sg.t] * k.dimf), gvt[sg.t].ensure(sizeof(double) * nt[sg.t] * k.dimf);
      QuadArgs qa;
      qa.lik = k.lik, qa.lik_param = k.param, qa.dimf = k.dimf, qa.Q = Q, qa.N = sg.n;
      qa.y = k.Y.d() + sg.r0;
      qa.yaux = k.Yaux.p ? k.Yaux.d() + sg.r0 : nullptr;
      qa.p = vp.d() + sg.off, qa.c = vc.d() + sg.off, qa.pt = vpt.d() + sg.off, qa.ct = vct.d() + sg.off;
      qa.ldn = ldn;
      std::memset(qa.w, 0, sizeof(qa.w)), std::memset(qa.w0, 0, sizeof(qa.w0)), std::memset(qa.kap, 0, sizeof(qa.kap));
      std::memset(qa.var, 0, sizeof(qa.var));
      for (int q = 0; q < Q; ++q) {
        qa.var[q] = h_var[q];
        for (int j = 0; j < k.dimf; ++j) {
          qa.w[q][j] = h_W[q * Df + k.d0 + j];
          qa.w0[q][j] = h_W0[q * Df + k.d0 + j];
          qa.kap[q][j] = h_kap[q * Df + k.d0 + j];
        }
      }
      qa.scale = h_bs[sg.t];
      qa.quirks = quirks;
      if (strict) qa.pg = vpg.d() + sg.off, qa.cg = vcg.d() + sg.off;
      qa.alpha = valpha.d() + sg.off, qa.beta = vbeta.d() + sg.off;      // rewritten with identical values
      qa.alpha0 = valpha0.d() + sg.off, qa.beta0 = vbeta0.d() + sg.off;
      qa.partials = quadpart.d();
      const long long within = sg.off - off[sg.t];
      qa.out_gm = gmt[sg.t].d() + within * k.dimf, qa.out_gv = gvt[sg.t].d() + within * k.dimf;
      launch_quad(qa, st);
    }
    tile.ensure(sizeof(double) * nmax * M);
    std::vector<double> hgv;
    size_t o1 = 0, o2 = 0;
    for (int q = 0; q < Q; ++q)
      for (int d = 0; d < Df; ++d) {
        const int t = f_index[d], j = d_index[d], J = tasks[t].dimf;
        const long long n = nt[t];
        if (o_kmn && n > 0) {
          launch_raw_kmn(a.d() + (long long)q * M, gmt[t].d(), gvt[t].d(), J, j, h_W[q * Df + d],
                         Pt.d() + (long long)q * ldn * M + off[t] * M, M, n, tile.d(), st);
          HIP_TRY(hipMemcpyAsync(o_kmn + o1, tile.p, sizeof(double) * n * M, hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
        }
        o1 += (size_t)n * M;
        if (o_kdiag && n > 0) {
          hgv.resize((size_t)n * J);
          HIP_TRY(hipMemcpy(hgv.data(), gvt[t].p, sizeof(double) * n * J, hipMemcpyDeviceToHost));
          for (long long i = 0; i < n; ++i) o_kdiag[o2 + i] = hgv[(size_t)i * J + j];
        }
        o2 += (size_t)n;
      }
  }

  // Natural-gradient update of q(u_q) = N(m_q, S_q) from the gradients of the last evaluation (the
  // north-star names it, the reference has none):  S^-1 <- S^-1 - 2 gamma dL/dS ;  S^-1 m <- S^-1 m + gamma (dL/dm -
  // 2 dL/dS m) ;  then m and L = chol(S) are recovered.  Requires the q(u) group in the last evaluation's mask.
  bool have_qu_grads = false;
  DevBuf ng_t1, ng_t2, ng_th, ng_mnew, ng_mq, ng_lflat;
  int* h_info2 = nullptr;   // page-locked: the two factorisations' info words of a natural-gradient step
  // Core of the natural-gradient step: leaves the new m_u ([M, Q], the layout of dmu) in ng_mq and the new packed Cholesky
  // factor in ng_lflat; throws HMOGP_E_NOT_PD (nothing modified) when the step leaves the positive-definite cone.  ONE host
  // synchronisation (the two info words) at the end; everything else is enqueued back to back on the engine's stream.
  void natgrad_core(double gamma, bool sync = true) {
    if (!evaluated || !have_qu_grads) throw EngineError{HMOGP_E_STATE, "natural-gradient step needs a finished evaluation with the q(u) group"};
    if (!(gamma > 0.0)) throw EngineError{HMOGP_E_INVALID, "bad natural-gradient arguments"};
    HIP_TRY(hipSetDevice(device));
    const long long Mtri = (long long)M * (M + 1) / 2;
    for (DevBuf* b : {&ng_t1, &ng_t2, &ng_th, &ng_mnew}) b->ensure(sizeof(double) * Q * M);
    ng_mq.ensure(sizeof(double) * M * Q), ng_lflat.ensure(sizeof(double) * Mtri * Q);
    if (!h_info2) HIP_TRY(hipHostMalloc((void**)&h_info2, sizeof(int) * 2 * HMOGP_MAXQ, hipHostMallocDefault));
    if (ng_pending) (void)qu_natgrad_status();                                       // (its info words sit where this step's will land)
    HIP_TRY(hipStreamWaitEvent(st, ev_join, 0));                                     // the q(u) tail of the evaluation (third stream)
    // Lambda = S^-1 - 2 gamma dL/dS is the new precision.  It is factorised REVERSED (rows and columns): J Lambda J = R R^T
    // gives Lambda = U U^T with U = J R J upper triangular, hence S_new = Lambda^-1 = U^-T U^-1 and L_new = U^-T = the
    // anti-transpose of R^-1 is the lower Cholesky factor of S_new (unique: positive diagonal) -- one factorisation and one
    // triangular inverse, no product R^-T R^-1 and no second factorisation.
    launch_natgrad_prec(Sqi.d(), dLdS.d(), gamma, G.d(), Q, M, true, st);
    launch_gemv_batched(Sqi.d(), dmu.d(), ng_t1.d(), Q, M, 1, Q, st);                // S^-1 m
    launch_gemv_batched(dLdS.d(), dmu.d(), ng_t2.d(), Q, M, 1, Q, st);               // dL/dS m
    launch_natgrad_theta1(ng_t1.d(), ng_t2.d(), gmu.d(), gamma, ng_th.d(), Q, M, st);
    launch_potrf_batched(G.d(), Q, M, dinfo.as<int>(), dscr.d(), st);                // J Lambda J = R R^T
    HIP_TRY(hipMemcpyAsync(h_info2, dinfo.p, sizeof(int) * Q, hipMemcpyDeviceToHost, st));
    // (speculative, like the K_uu chain of an evaluation: a failed factorisation makes the launches below no-ops on garbage
    //  that is never committed)
    launch_trtri_batched(G.d(), tmpA.d(), tmpB.d(), Q, M, st);                       // R^-1
    launch_antitranspose(tmpA.d(), GSK.d(), Q, M, st);                               // L_new[i][j] = R^-1[M-1-j][M-1-i]
    launch_gemv_t_batched(GSK.d(), ng_th.d(), ng_t1.d(), Q, M, st);                  // L^T theta1
    launch_gemv_batched(GSK.d(), ng_t1.d(), ng_mnew.d(), Q, M, M, 1, st);            // m_new = S_new theta1 = L (L^T theta1)
    launch_pack_tril(GSK.d(), ng_lflat.d(), Q, M, 1.0, st);
    launch_scatter_mq(ng_mnew.d(), ng_mq.d(), Q, M, st);
    if (!sync) return;               // hmogp_qu_natgrad_async: the commit is decided on the device, the host looks later
    HIP_TRY(hipStreamSynchronize(st));
    // (a failed step has only written scratch -- G, GSK, tmpA, tmpB -- none of which is an input of the step: the caller
    //  may retry with a smaller gamma straight away, no new evaluation needed)
    for (int q = 0; q < Q; ++q)
      if (h_info2[q] != 0) throw EngineError{HMOGP_E_NOT_PD, "natural-gradient step leaves the positive-definite cone (reduce gamma)"};
    evaluated = false;  // q(u) moves on: posterior / predict / another step need a fresh evaluation
  }
  // Natural-gradient update of q(u_q) = N(m_q, S_q) from the gradients of the last evaluation (the
  // north-star names it, the reference has none):  S^-1 <- S^-1 - 2 gamma dL/dS ;  S^-1 m <- S^-1 m + gamma (dL/dm -
  // 2 dL/dS m) ;  then m and L = chol(S) are recovered.  Requires the q(u) group in the last evaluation's mask.
  void natgrad_step(double gamma, double* m_out, double* L_flat_out) {
    if (!m_out || !L_flat_out) throw EngineError{HMOGP_E_INVALID, "bad natural-gradient arguments"};
    natgrad_core(gamma);
    const long long Mtri = (long long)M * (M + 1) / 2;
    HIP_TRY(hipMemcpyAsync(m_out, ng_mq.p, sizeof(double) * M * Q, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(L_flat_out, ng_lflat.p, sizeof(double) * Mtri * Q, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  // The same step on the DEVICE-RESIDENT q(u) (hmogp_qu_load): m_u / L_flat are updated in place in HBM, nothing but the two
  // info words crosses PCIe -- the natural-gradient SVI loop (E-steps) without moving 2 x 12.6 MB per iteration.
  void qu_natgrad(double gamma) {
    if (!qu_resident) throw EngineError{HMOGP_E_STATE, "no resident q(u) (hmogp_qu_load)"};
    natgrad_core(gamma);
    const long long Mtri = (long long)M * (M + 1) / 2;
    HIP_TRY(hipMemcpyAsync(dmu.p, ng_mq.p, sizeof(double) * M * Q, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(dLflat.p, ng_lflat.p, sizeof(double) * Mtri * Q, hipMemcpyDeviceToDevice, st));
    // (no synchronisation: the next evaluation reads q(u) on streams ordered behind this one -- see upload_params)
    HIP_TRY(hipEventRecord(ev_qu, st));
  }

  // [r5, ABI v7] hmogp_qu_natgrad without the host synchronisation: the commit into the resident q(u) is conditional ON THE DEVICE
  // (commit_if_ok_kernel reads the factorisation's info words), the call returns with everything enqueued, and the caller goes
  // straight on to the next evaluation -- whose parameter upload, pool staging and K_uf construction (second stream) then run BESIDE
  // this step's latency-bound factorisation chain instead of behind a host round trip.  hmogp_qu_natgrad_status waits and reports.
  hipEvent_t ev_ng = nullptr;
  bool ng_pending = false, ng_last_taken = true;
  void qu_natgrad_async(double gamma) {
    if (!qu_resident) throw EngineError{HMOGP_E_STATE, "no resident q(u) (hmogp_qu_load)"};
    if (ng_pending) throw EngineError{HMOGP_E_STATE, "a natural-gradient step is pending (hmogp_qu_natgrad_status)"};
    natgrad_core(gamma, false);
    const long long Mtri = (long long)M * (M + 1) / 2;
    launch_commit_if_ok(dinfo.as<int>(), Q, ng_mq.d(), dmu.d(), (long long)M * Q, ng_lflat.d(), dLflat.d(), Mtri * Q, st);
    HIP_TRY(hipEventRecord(ev_qu, st));
    if (!ev_ng) HIP_TRY(hipEventCreateWithFlags(&ev_ng, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev_ng, st));
    ng_pending = true;
    evaluated = false;               // q(u) (probably) moves on: posterior / predict / another step need a fresh evaluation
  }
  int qu_natgrad_status() {
    if (ng_pending) {
      HIP_TRY(hipSetDevice(device));
      HIP_TRY(hipEventSynchronize(ev_ng));
      ng_pending = false;
      ng_last_taken = true;
      for (int q = 0; q < Q; ++q) ng_last_taken = ng_last_taken && h_info2[q] == 0;
    }
    return ng_last_taken ? 1 : 0;
  }

  void predict_f(const double* Xnew, long long Nnew, double* m, double* v) {
    if (!evaluated && !began) throw EngineError{HMOGP_E_STATE, "no evaluation to predict from"};
    if (Nnew < 0 || (Nnew > 0 && (!Xnew || !m || !v))) throw EngineError{HMOGP_E_INVALID, "bad predict arguments"};
    HIP_TRY(hipSetDevice(device));
    const long long MM = (long long)M * M;
    const int ldz = Q * P;
    ensure_workspace(std::min(chunk, std::max<long long>(Nnew, 1)));
    ensure_strict_workspace();           // (predictions follow the mode of the evaluation they are taken from)
    const long long ldn = ws_rows;
    DevBuf dX, dm, dv;
    dX.ensure(sizeof(double) * ldn * P), dm.ensure(sizeof(double) * ldn * Df), dv.ensure(sizeof(double) * ldn * Df);
    for (long long r0 = 0; r0 < Nnew; r0 += ldn) {
      const long long n = std::min(ldn, Nnew - r0);
      HIP_TRY(hipMemcpyAsync(dX.p, Xnew + r0 * P, sizeof(double) * n * P, hipMemcpyHostToDevice, st));
      if (strict) {
        RbfBatch rbt;
        rbt.nq = Q, rbt.var = dvar.d(), rbt.ell = dell.d(), rbt.sZ = P, rbt.sK = ldn * M;
        launch_rbf(dX.d(), P, n, P, dZ.d(), ldz, M, 0.0, 1.0, Kh.d(), false, st, nullptr, true, &rbt);
        strict_forward(n, dX.d(), false, false);
      }
      for (int q = 0; q < Q && !strict; ++q) {
        double* kh = Kh.d() + (long long)q * ldn * M;
        double* pt = Pt.d() + (long long)q * ldn * M;
        launch_rbf(dX.d(), P, n, P, dZ.d() + q * P, ldz, M, h_var[q], h_ell[q], kh, false, st, nullptr, false);
        GemmArgs g;
        g.A = kh, g.lda = M, g.a_kmajor = 0;
        g.B = Ctri.d() + q * MM, g.ldb = M, g.b_kmajor = 1, g.b_tri = 1;  // variances only: triangular fold of C
        g.C = pt, g.ldc = M;
        g.M = (int)n, g.N = M, g.K = M;
        g.role = 1;
        g.fs_part = fwdpart.d(), g.fs_a = a.d() + (long long)q * M, g.fs_x = dX.d(), g.fs_z = dZ.d() + q * P;
        g.fs_ldz = ldz, g.fs_P = P, g.fs_hyper = 0, g.fs_ell = dell.d() + q;
        g.store_c = 0;
        const int nparts = launch_gemm_rowpass_or_general(g, st);
        launch_combine_parts(fwdpart.d(), nparts * ((M + 127) / 128), n, vp.d() + q * ldn, vc.d() + q * ldn, nullptr, nullptr, st);
      }
      launch_qf_combine(vp.d(), vc.d(), ldn, n, Q, Df, dW.d(), dkap.d(), dvar.d(), dm.d(), dv.d(), st);
      HIP_TRY(hipMemcpyAsync(m + r0 * Df, dm.p, sizeof(double) * n * Df, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(v + r0 * Df, dv.p, sizeof(double) * n * Df, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
  }
};

// =================================================================================================== C ABI
namespace {

template <class F>
int guarded(hmogp_engine* h, F&& f) {
  std::string* err = h ? &h->err : &g_create_error;
  try {
    f();
    return HMOGP_OK;
  } catch (const EngineError& e) {
    *err = e.msg;
    return e.code;
  } catch (const HipError& e) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d in %s", (int)e.code, hipGetErrorString(e.code), e.file, e.line,
                  e.what);
    *err = buf;
    return HMOGP_E_NO_DEVICE;
  } catch (const std::exception& e) {
    *err = e.what();
    return HMOGP_E_INVALID;
  }
}

void need_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    throw EngineError{HMOGP_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)"};
  if (device < 0 || device >= ndev) throw EngineError{HMOGP_E_NO_DEVICE, "HIP device ordinal out of range"};
  HIP_TRY(hipSetDevice(device));
}

}  // namespace

extern "C" {

int hmogp_abi_version(void) { return HMOGP_ABI_VERSION; }

void* hmogp_host_alloc(uint64_t bytes) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || bytes == 0) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void hmogp_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int hmogp_create(const hmogp_config* cfg, hmogp_handle* out) {
  if (!out) return HMOGP_E_INVALID;
  *out = nullptr;
  hmogp_engine* e = nullptr;
  int rc = guarded(nullptr, [&] {
    e = new hmogp_engine();
    e->init(cfg);
  });
  if (rc != HMOGP_OK) {
    delete e;
    return rc;
  }
  *out = e;
  return HMOGP_OK;
}

void hmogp_destroy(hmogp_handle h) { delete h; }

const char* hmogp_last_error(hmogp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int hmogp_set_task_data(hmogp_handle h, int32_t t, const double* X, const double* Y, int64_t N) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->set_task_data(t, X, Y, N); });
}

int hmogp_step_begin(hmogp_handle h, const hmogp_params* p) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->begin(p); });
}

int hmogp_stats_buffer(hmogp_handle h, void** device_ptr, int64_t* count) {
  if (!h || !device_ptr || !count) return HMOGP_E_INVALID;
  *device_ptr = h->stats.p;
  *count = h->nstats;
  return HMOGP_OK;
}

int hmogp_stats_read(hmogp_handle h, double* host) {
  if (!h || !host) return HMOGP_E_INVALID;
  return guarded(h, [&] {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(host, h->stats.p, sizeof(double) * h->nstats, hipMemcpyDeviceToHost));
  });
}

int hmogp_stats_write(hmogp_handle h, const double* host) {
  if (!h || !host) return HMOGP_E_INVALID;
  return guarded(h, [&] {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(h->stats.p, host, sizeof(double) * h->nstats, hipMemcpyHostToDevice));
  });
}

int hmogp_wire_buffer(hmogp_handle h, void** device_ptr, int64_t* count) {
  if (!h || !device_ptr || !count) return HMOGP_E_INVALID;
  return guarded(h, [&] {
    HIP_TRY(hipSetDevice(h->device));
    h->wire.ensure(sizeof(double) * h->nwire, true);
    *device_ptr = h->wire.p;
    *count = h->nwire;
  });
}

int hmogp_wire_pack(hmogp_handle h) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->wire_copy(0); });
}

int hmogp_wire_unpack(hmogp_handle h) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->wire_copy(1); });
}

int hmogp_wire_read(hmogp_handle h, double* host) {
  if (!h || !host) return HMOGP_E_INVALID;
  return guarded(h, [&] {
    HIP_TRY(hipSetDevice(h->device));
    h->wire.ensure(sizeof(double) * h->nwire, true);
    HIP_TRY(hipMemcpy(host, h->wire.p, sizeof(double) * h->nwire, hipMemcpyDeviceToHost));
  });
}

int hmogp_wire_write(hmogp_handle h, const double* host) {
  if (!h || !host) return HMOGP_E_INVALID;
  return guarded(h, [&] {
    HIP_TRY(hipSetDevice(h->device));
    h->wire.ensure(sizeof(double) * h->nwire, true);
    HIP_TRY(hipMemcpy(h->wire.p, host, sizeof(double) * h->nwire, hipMemcpyHostToDevice));
  });
}

int hmogp_comm_available(void) { return rccl().ok() ? 1 : 0; }

int hmogp_comm_unique_id(void* id128) {
  if (!id128) return HMOGP_E_INVALID;
  return guarded(nullptr, [&] {
    RcclApi& r = rccl();
    if (!r.ok()) throw EngineError{HMOGP_E_COMM, "librccl not available: " + r.why};
    static_assert(sizeof(hm_nccl::UniqueId) == HMOGP_COMM_ID_BYTES, "ncclUniqueId size");
    hm_nccl::UniqueId uid;
    RCCL_TRY(r.getUniqueId(&uid));
    std::memcpy(id128, &uid, sizeof uid);
  });
}

int hmogp_comm_init(hmogp_handle h, int32_t nranks, int32_t rank, const void* id128) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->comm_init(nranks, rank, id128); });
}

int hmogp_comm_destroy(hmogp_handle h) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->comm_destroy(); });
}

int hmogp_comm_info(hmogp_handle h, int32_t* nranks, int32_t* rank) {
  if (!h) return HMOGP_E_INVALID;
  if (nranks) *nranks = h->comm ? h->comm_ranks : 0;
  if (rank) *rank = h->comm ? h->comm_rank : -1;
  return HMOGP_OK;
}

int hmogp_step_exchange(hmogp_handle h) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->exchange(); });
}

int hmogp_step_finish(hmogp_handle h, hmogp_outputs* out) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->finish(out); });
}

int hmogp_elbo_grad(hmogp_handle h, const hmogp_params* p, hmogp_outputs* out) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] {       // single device, NEVER a collective -- also with a communicator attached (debug / parity calls)
    static const bool stamps = getenv("HMOGP_HOST_STAMPS") != nullptr;   // host-side timeline of the call (stderr; debugging)
    static std::chrono::steady_clock::time_point last_ret;
    const auto t_in = std::chrono::steady_clock::now();
    try {
      h->pending_warm.clear();
      if (h->graphs_broken || !h->graph_step(p, out)) {
        h->begin(p, false);
        const auto t_b = std::chrono::steady_clock::now();
        h->finish_enqueue(out);
        const auto t_e = std::chrono::steady_clock::now();
        if (stamps) HIP_TRY(hipStreamSynchronize(h->st));
        const auto t_s = std::chrono::steady_clock::now();
        h->finish_tail(out);
        h->mark_warm();
        if (stamps) {
          const auto t_r = std::chrono::steady_clock::now();
          auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
          std::fprintf(stderr, "[hmogp host] outside %.0f us | begin (enqueue) %.0f | finish enqueue %.0f | wait %.0f | tail %.0f\n",
                       us(last_ret, t_in), us(t_in, t_b), us(t_b, t_e), us(t_e, t_s), us(t_s, t_r));
          last_ret = t_r;
        }
      }
    } catch (const hmogp_engine::RetryRegular&) {   // small-model path: a latent needs the jitter ladder
      h->small_veto = true;
      try {
        h->begin(p, false);
        h->finish(out);
      } catch (...) {
        h->small_veto = false;
        throw;
      }
      h->small_veto = false;
    }
  });
}

int hmogp_elbo_grad_sharded(hmogp_handle h, const hmogp_params* p, hmogp_outputs* out) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] {
    if (!h->comm) throw EngineError{HMOGP_E_STATE, "hmogp_elbo_grad_sharded without a communicator (hmogp_comm_init)"};
    try {
      h->begin(p, false, true);
      h->exchange();            // the one collective of the path, enqueued between the two halves on the engine's stream
    } catch (...) {
      h->comm_abort();          // this rank cannot contribute: the peers must fail, not hang
      throw;
    }
    h->finish(out);
  });
}

int hmogp_posterior_u(hmogp_handle h, double* woodbury_vector, double* woodbury_inv) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->posterior_u(woodbury_vector, woodbury_inv); });
}

int hmogp_qu_load(hmogp_handle h, const double* m_u, const double* L_flat) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->qu_load(m_u, L_flat); });
}

int hmogp_qu_read(hmogp_handle h, double* m_u, double* L_flat) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->qu_read(m_u, L_flat); });
}

int hmogp_qu_adadelta(hmogp_handle h, int32_t phase, double step_rate, double momentum, double decay, double one_minus_decay,
                      double offset) {
  if (!h || (phase != 0 && phase != 1)) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->qu_adadelta(phase, step_rate, momentum, decay, one_minus_decay, offset); });
}

int hmogp_debug_raw_grads(hmogp_handle h, double* dL_dKmm, double* dL_dKmn, double* dL_dKdiag) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->debug_raw(dL_dKmm, dL_dKmn, dL_dKdiag); });
}

int hmogp_natgrad_step(hmogp_handle h, double gamma, double* m_u_new, double* L_flat_new) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->natgrad_step(gamma, m_u_new, L_flat_new); });
}

int hmogp_graph_stats(hmogp_handle h, int64_t* captures, int64_t* replays) {
  if (!h) return HMOGP_E_INVALID;
  if (captures) *captures = h->graph_captures;
  if (replays) *replays = h->graph_replays;
  return HMOGP_OK;
}

int hmogp_qu_natgrad_async(hmogp_handle h, double gamma) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->qu_natgrad_async(gamma); });
}
int hmogp_qu_natgrad_status(hmogp_handle h, int32_t* taken) {
  if (!h || !taken) return HMOGP_E_INVALID;
  return guarded(h, [&] { *taken = h->qu_natgrad_status(); });
}
int hmogp_qu_natgrad(hmogp_handle h, double gamma) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->qu_natgrad(gamma); });
}

int hmogp_predict_f(hmogp_handle h, const double* Xnew, int64_t Nnew, double* m, double* v) {
  if (!h) return HMOGP_E_INVALID;
  return guarded(h, [&] { h->predict_f(Xnew, Nnew, m, v); });
}

int hmogp_last_timings(hmogp_handle h, double* out_ms, int64_t* launches) {
  if (!h || !out_ms) return HMOGP_E_INVALID;
  for (int c = 0; c < NCAT; ++c) {
    out_ms[c] = h->ms[c];
    if (launches) launches[c] = h->launches[c];
  }
  return HMOGP_OK;
}

// ---- building blocks -----------------------------------------------------------------------------------
int hmogp_rbf_cross_cov(int32_t device, const double* X, int64_t N, const double* Z, int32_t M, int32_t P, double variance,
                        double lengthscale, double* K) {
  return hmogp_rbf_cross_cov_ex(device, X, N, Z, M, P, variance, lengthscale, 1, K);
}

int hmogp_rbf_cross_cov_ex(int32_t device, const double* X, int64_t N, const double* Z, int32_t M, int32_t P, double variance,
                           double lengthscale, int32_t exact, double* K) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (N <= 0 || M <= 0 || !X || !Z || !K) throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    DevBuf dX, dZ, dK;
    dX.ensure(sizeof(double) * N * P), dZ.ensure(sizeof(double) * M * P), dK.ensure(sizeof(double) * N * M);
    HIP_TRY(hipMemcpy(dX.p, X, sizeof(double) * N * P, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dZ.p, Z, sizeof(double) * M * P, hipMemcpyHostToDevice));
    launch_rbf(dX.d(), P, N, P, dZ.d(), P, M, variance, lengthscale, dK.d(), false, nullptr, nullptr, exact != 0);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(K, dK.p, sizeof(double) * N * M, hipMemcpyDeviceToHost));
  });
}

int hmogp_jitchol_inv(int32_t device, const double* A, int32_t Q, int32_t M, const int32_t* forced_rung, double* L,
                      double* Ainv, int32_t* rung) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (Q <= 0 || M <= 0 || !A) throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    const long long MM = (long long)M * M;
    DevBuf dA, dL, dLi, dT, dO, info, jit, scr;
    for (DevBuf* b : {&dA, &dL, &dLi, &dT, &dO}) b->ensure(sizeof(double) * MM * Q, true);
    info.ensure(sizeof(int) * Q), jit.ensure(sizeof(double) * Q), scr.ensure(sizeof(double) * Q * M * M);
    HIP_TRY(hipMemcpy(dA.p, A, sizeof(double) * MM * Q, hipMemcpyHostToDevice));
    std::vector<double> dmean(Q);
    std::vector<int> r(Q);
    for (int q = 0; q < Q; ++q) {
      double s = 0.0;
      for (int i = 0; i < M; ++i) s += A[q * MM + (long long)i * M + i];
      dmean[q] = s / M;
      r[q] = forced_rung ? forced_rung[q] : -2;
    }
    jitchol_batched(dA.d(), dL.d(), Q, M, dmean.data(), r.data(), info.as<int>(), jit.d(), scr.d(), nullptr);
    if (rung) std::copy(r.begin(), r.end(), rung);
    if (L) HIP_TRY(hipMemcpy(L, dL.p, sizeof(double) * MM * Q, hipMemcpyDeviceToHost));
    if (Ainv) {
      launch_trtri_batched(dL.d(), dLi.d(), dT.d(), Q, M, nullptr);
      launch_ltl_batched(dLi.d(), dO.d(), Q, M, nullptr);
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(Ainv, dO.p, sizeof(double) * MM * Q, hipMemcpyDeviceToHost));
    }
  });
}

int hmogp_potri(int32_t device, const double* L, int32_t Q, int32_t M, double* Sinv) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (Q <= 0 || M <= 0 || !L || !Sinv) throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    const long long MM = (long long)M * M;
    DevBuf dL, dLi, dT, dO;
    for (DevBuf* b : {&dL, &dLi, &dT, &dO}) b->ensure(sizeof(double) * MM * Q, true);
    HIP_TRY(hipMemcpy(dL.p, L, sizeof(double) * MM * Q, hipMemcpyHostToDevice));
    launch_trtri_batched(dL.d(), dLi.d(), dT.d(), Q, M, nullptr);
    launch_ltl_batched(dLi.d(), dO.d(), Q, M, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(Sinv, dO.p, sizeof(double) * MM * Q, hipMemcpyDeviceToHost));
  });
}

int hmogp_potrs_rows(int32_t device, const double* L, int32_t M, const double* B, int64_t n, double* out) {
  return guarded(nullptr, [&] {
    if (!L || !B || !out || M < 1 || n < 0) throw EngineError{HMOGP_E_INVALID, "bad potrs arguments"};
    need_device(device);
    HIP_TRY(hipSetDevice(device));
    DevBuf dL, dV;
    dL.ensure(sizeof(double) * M * M), dV.ensure(sizeof(double) * std::max<long long>(1, n) * M);
    HIP_TRY(hipMemcpy(dL.p, L, sizeof(double) * M * M, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dV.p, B, sizeof(double) * n * M, hipMemcpyHostToDevice));
    potrs_rows_inplace(dV.d(), n * M, dL.d(), (long long)M * M, M, n, 1, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dV.p, sizeof(double) * n * M, hipMemcpyDeviceToHost));
  });
}

int hmogp_gemm_f64(int32_t device, int32_t transA, int32_t transB, int32_t M, int32_t N, int32_t K, double alpha,
                   const double* A, int32_t lda, const double* B, int32_t ldb, double beta, double* C, int32_t ldc) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    const long long na = (long long)(transA ? K : M) * lda, nb = (long long)(transB ? N : K) * ldb, nc = (long long)M * ldc;
    DevBuf dA, dB, dC;
    dA.ensure(sizeof(double) * na), dB.ensure(sizeof(double) * nb), dC.ensure(sizeof(double) * nc);
    HIP_TRY(hipMemcpy(dA.p, A, sizeof(double) * na, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dB.p, B, sizeof(double) * nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dC.p, C, sizeof(double) * nc, hipMemcpyHostToDevice));
    GemmArgs g;
    g.A = dA.d(), g.B = dB.d(), g.C = dC.d();
    g.M = M, g.N = N, g.K = K;
    g.lda = lda, g.ldb = ldb, g.ldc = ldc;
    g.a_kmajor = transA ? 1 : 0;  // op(A) = A^T: A stored [k][i]
    g.b_kmajor = transB ? 0 : 1;  // op(B) = B^T: B stored [j][k]
    g.alpha = alpha, g.beta = beta;
    launch_gemm_f64(g, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(C, dC.p, sizeof(double) * nc, hipMemcpyDeviceToHost));
  });
}

int hmogp_var_exp(int32_t device, int32_t lik_id, double lik_param, int64_t N, const double* y, const double* m,
                  const double* v, double* ve, double* dm, double* dv) {
  return hmogp_var_exp_ex(device, lik_id, lik_param, HMOGP_QUIRKS_REFERENCE, N, y, m, v, ve, dm, dv);
}

int hmogp_var_exp_ex(int32_t device, int32_t lik_id, double lik_param, uint32_t quirks, int64_t N, const double* y,
                     const double* m, const double* v, double* ve, double* dm, double* dv) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (lik_id == HMOGP_LIK_GAUSSIAN && !(lik_param > 0.0)) lik_param = 0.5;
    const int J = lik_dimf(lik_id, lik_param);
    if (J < 1 || J > HMOGP_MAXJ || N <= 0 || !y || !m || !v || !ve || !dm || !dv)
      throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    DevBuf dy, dmm, dvv, dve, ddm, ddv;
    dy.ensure(sizeof(double) * N), dve.ensure(sizeof(double) * N);
    for (DevBuf* b : {&dmm, &dvv, &ddm, &ddv}) b->ensure(sizeof(double) * N * J);
    HIP_TRY(hipMemcpy(dy.p, y, sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dmm.p, m, sizeof(double) * N * J, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dvv.p, v, sizeof(double) * N * J, hipMemcpyHostToDevice));
    launch_var_exp(lik_id, J, lik_param, N, dy.d(), dmm.d(), dvv.d(), dve.d(), ddm.d(), ddv.d(), nullptr, quirks);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(ve, dve.p, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dm, ddm.p, sizeof(double) * N * J, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dv, ddv.p, sizeof(double) * N * J, hipMemcpyDeviceToHost));
  });
}

int hmogp_predictive(int32_t device, int32_t lik_id, double lik_param, int32_t gh_T, int64_t N, const double* m,
                     const double* v, double* mean, double* var) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (lik_id == HMOGP_LIK_GAUSSIAN && !(lik_param > 0.0)) lik_param = 0.5;
    const int J = lik_dimf(lik_id, lik_param);
    const int Jp = (lik_id == HMOGP_LIK_CATEGORICAL) ? J : 1;  // dim_p of the reference's get_metadata()
    if (gh_T == 0) gh_T = (lik_id == HMOGP_LIK_CATEGORICAL) ? 10 : 20;
    if (J < 1 || J > HMOGP_MAXJ || N <= 0 || !m || !v || !mean || !var || (gh_T != 10 && gh_T != 20))
      throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    DevBuf dm, dv, om, ov;
    dm.ensure(sizeof(double) * N * J), dv.ensure(sizeof(double) * N * J);
    om.ensure(sizeof(double) * N * Jp), ov.ensure(sizeof(double) * N * Jp);
    HIP_TRY(hipMemcpy(dm.p, m, sizeof(double) * N * J, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv.p, v, sizeof(double) * N * J, hipMemcpyHostToDevice));
    launch_predictive(lik_id, J, Jp, lik_param, gh_T, N, dm.d(), dv.d(), om.d(), ov.d(), nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(mean, om.p, sizeof(double) * N * Jp, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(var, ov.p, sizeof(double) * N * Jp, hipMemcpyDeviceToHost));
  });
}

int hmogp_log_predictive(int32_t device, int32_t lik_id, double lik_param, int64_t N, int32_t num_samples, uint64_t seed,
                         const double* y, const double* m, const double* v, double* log_pred) {
  return guarded(nullptr, [&] {
    need_device(device);
    const int J = lik_dimf(lik_id, lik_param);
    if (J < 1 || J > HMOGP_MAXJ || N <= 0 || num_samples < 1 || !y || !m || !v || !log_pred)
      throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    if (lik_id == HMOGP_LIK_GAMMA || lik_id == HMOGP_LIK_BETA)
      throw EngineError{HMOGP_E_INVALID, "the reference defines no log_predictive for Gamma / Beta"};
    DevBuf dy, dm, dv, dout;
    dy.ensure(sizeof(double) * N), dout.ensure(sizeof(double) * N);
    dm.ensure(sizeof(double) * N * J), dv.ensure(sizeof(double) * N * J);
    HIP_TRY(hipMemcpy(dy.p, y, sizeof(double) * N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dm.p, m, sizeof(double) * N * J, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv.p, v, sizeof(double) * N * J, hipMemcpyHostToDevice));
    launch_log_predictive(lik_id, J, lik_param, N, num_samples, seed, dy.d(), dm.d(), dv.d(), dout.d(), nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(log_pred, dout.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  });
}

int hmogp_sample(int32_t device, int32_t lik_id, double lik_param, int64_t N, uint64_t seed, const double* F, double* Y) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (lik_id == HMOGP_LIK_GAUSSIAN && !(lik_param > 0.0)) lik_param = 0.5;
    const int J = lik_dimf(lik_id, lik_param);
    if (J < 1 || J > HMOGP_MAXJ || N <= 0 || !F || !Y) throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    DevBuf dF, dY;
    dF.ensure(sizeof(double) * N * J), dY.ensure(sizeof(double) * N);
    HIP_TRY(hipMemcpy(dF.p, F, sizeof(double) * N * J, hipMemcpyHostToDevice));
    launch_sample(lik_id, J, lik_param, N, seed, dF.d(), dY.d(), nullptr);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(Y, dY.p, sizeof(double) * N, hipMemcpyDeviceToHost));
  });
}

int hmogp_bench_contraction(int32_t device, int32_t role, int64_t n, int32_t M, int32_t iters, double* avg_ms) {
  return guarded(nullptr, [&] {
    need_device(device);
    if (n <= 0 || M <= 0 || iters <= 0 || !avg_ms || role < 1 || role > 6) throw EngineError{HMOGP_E_INVALID, "bad arguments"};
    const long long MM = (long long)M * M;
    if (role == 5) {  // K_uf construction alone: the launch shape of the row pass (3 latents batched, P = 1, hot-path variant)
      const int Qb = 3;
      DevBuf X, Z, K, var, ell;
      X.ensure(sizeof(double) * n), Z.ensure(sizeof(double) * M * Qb), K.ensure(sizeof(double) * n * M * Qb);
      var.ensure(sizeof(double) * Qb), ell.ensure(sizeof(double) * Qb);
      std::vector<double> hx((size_t)n), hz((size_t)M * Qb), hv(Qb, 0.5), hl(Qb);
      for (long long i = 0; i < n; ++i) hx[(size_t)i] = (double)i / (double)n;
      for (int m = 0; m < M; ++m)
        for (int q = 0; q < Qb; ++q) hz[(size_t)m * Qb + q] = (double)m / (double)std::max(1, M - 1);
      for (int q = 0; q < Qb; ++q) hl[q] = (0.8 + 0.25 * q) / (double)std::max(1, M - 1);
      HIP_TRY(hipMemcpy(X.p, hx.data(), sizeof(double) * n, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(Z.p, hz.data(), sizeof(double) * M * Qb, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(var.p, hv.data(), sizeof(double) * Qb, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(ell.p, hl.data(), sizeof(double) * Qb, hipMemcpyHostToDevice));
      RbfBatch rbt;
      rbt.nq = Qb, rbt.var = var.d(), rbt.ell = ell.d(), rbt.sZ = 1, rbt.sK = n * (long long)M;
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      launch_rbf(X.d(), 1, n, 1, Z.d(), Qb, M, 0.0, 1.0, K.d(), false, nullptr, nullptr, false, &rbt);
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) launch_rbf(X.d(), 1, n, 1, Z.d(), Qb, M, 0.0, 1.0, K.d(), false, nullptr, nullptr, false, &rbt);
      HIP_TRY(hipEventRecord(e1, nullptr));
      HIP_TRY(hipEventSynchronize(e1));
      float msf = 0.f;
      HIP_TRY(hipEventElapsedTime(&msf, e0, e1));
      *avg_ms = msf / iters;
      (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
      return;
    }
    DevBuf A, B, Cc, beta, slabs;
    A.ensure(sizeof(double) * n * M), B.ensure(sizeof(double) * MM), Cc.ensure(sizeof(double) * std::max<long long>(n * M, MM));
    beta.ensure(sizeof(double) * n), slabs.ensure(sizeof(double) * MM * gram_ksplit(n, M), true);
    DevBuf part, ell;   // roles 3 / 4: forward contraction with the fused row-statistics epilogue (with / without P~ store)
    part.ensure(sizeof(double) * 4 * FWD_PARTS * ((M + 127) / 128) * n, true), ell.ensure(sizeof(double), true);
    { const double one = 1.0; HIP_TRY(hipMemcpy(ell.p, &one, sizeof(double), hipMemcpyHostToDevice)); }
    std::vector<double> h((size_t)std::max<long long>(n * M, MM));
    unsigned long long s = 88172645463325252ULL;   // xorshift: full-range random operands (DVFS-realistic, guide rule 25)
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
    for (auto& v : h) v = rnd();
    HIP_TRY(hipMemcpy(A.p, h.data(), sizeof(double) * n * M, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(B.p, h.data(), sizeof(double) * MM, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(beta.p, h.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto once = [&] {
      GemmArgs g;
      if (role == 6) {   // diagnostic: the weighted Gram over ALL tiles (no lower-only handling), no slab reduction
        const int ksplit = gram_ksplit(n, M);
        g.A = A.d(), g.lda = M, g.a_kmajor = 1;
        g.B = A.d(), g.ldb = M, g.b_kmajor = 1;
        g.kscale = beta.d();
        g.C = slabs.d(), g.ldc = M;
        g.M = g.N = M, g.K = (int)n;
        g.lower_only = 0, g.ksplit = ksplit, g.sSplit = MM, g.role = 2;
        launch_gemm_rowpass_or_general(g, nullptr);
      } else if (role != 2) {
        if (role >= 3) {
          g.fs_part = part.d(), g.fs_a = beta.d(), g.fs_x = beta.d(), g.fs_z = B.d(), g.fs_ldz = 1, g.fs_P = 1;
          g.fs_hyper = 1, g.fs_ell = ell.d(), g.store_c = role == 3 ? 1 : 0;
        }
        g.A = A.d(), g.lda = M, g.a_kmajor = 0;
        g.B = B.d(), g.ldb = M, g.b_kmajor = 1;
        g.C = Cc.d(), g.ldc = M;
        g.M = (int)n, g.N = M, g.K = M;
        g.role = 1;
        launch_gemm_rowpass_or_general(g, nullptr);
      } else {
        const int ksplit = gram_ksplit(n, M);
        g.A = A.d(), g.lda = M, g.a_kmajor = 1;
        g.B = A.d(), g.ldb = M, g.b_kmajor = 1;
        g.kscale = beta.d();
        g.C = slabs.d(), g.ldc = M;
        g.M = g.N = M, g.K = (int)n;
        g.lower_only = 1, g.ksplit = ksplit, g.sSplit = MM, g.role = 2;
        launch_gemm_rowpass_or_general(g, nullptr);
        launch_reduce_slabs_lower(slabs.d(), ksplit, M, Cc.d(), true, nullptr);
      }
    };
    once();
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) once();
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  });
}

}  // extern "C"
