// abi.hip -- the C ABI of include/hetmogp_hip.h: argument checks, error mapping (guarded), the engine entry points and the
// stand-alone building blocks.  Split out of engine.hip in round 6 (no behaviour change).
#include "engine_impl.h"

// =================================================================================================== C ABI
namespace {


thread_local std::string g_create_error;

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
    // [r6] with scratch for the mirrored factor and the pivots' reciprocals the call takes the one-launch-per-block kernels where the
    // shape allows it (M a multiple of 128, n >= 1024: trsm_panel.hip), else the round-5 path: the building block runs what the
    // strict mode runs (tests/test_gpu_strict.py::test_potrs_rows_vs_lapack at both kinds of shape)
    DevBuf dS, dR;
    dS.ensure(sizeof(double) * M * M), dR.ensure(sizeof(double) * M);
    potrs_rows_inplace(dV.d(), n * M, dL.d(), (long long)M * M, M, n, 1, nullptr, dS.d(), nullptr, dR.d());
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
