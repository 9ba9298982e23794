// engine_comm.hip -- the exchange step of a row-sharded run: librccl resolved at run time, one ncclAllReduce of the wire format per step.
// Split out of engine.hip in round 6 (no behaviour change); declarations: engine_impl.h.
#include "engine_impl.h"

namespace hmogp_detail {
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

}  // namespace hmogp_detail

void hmogp_engine::comm_init(int nranks, int rank, const void* id) {
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

void hmogp_engine::comm_destroy() {
  if (!comm) return;
  (void)hipSetDevice(device);
  (void)hipStreamSynchronize(st);
  (void)rccl().commDestroy(comm);
  comm = nullptr, comm_ranks = 1, comm_rank = 0;
}

void hmogp_engine::comm_abort() {
  if (!comm) return;
  RcclApi& r = rccl();
  if (r.commAbort) (void)r.commAbort(comm);
  else (void)r.commDestroy(comm);
  comm = nullptr, comm_ranks = 1, comm_rank = 0;
}

void hmogp_engine::wait_exchanged() {
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

void hmogp_engine::exchange() {
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

void hmogp_engine::wire_copy(int dir) {
  if (!began) throw EngineError{HMOGP_E_STATE, "wire pack / unpack outside hmogp_step_begin .. hmogp_step_finish"};
  HIP_TRY(hipSetDevice(device));
  wire.ensure(sizeof(double) * nwire, true);
  launch_wire_copy(stats.d(), wire.d(), NG, Q, M, per_q, dir, st);
  HIP_TRY(hipStreamSynchronize(st));
}
