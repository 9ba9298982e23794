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
// [r6] This file keeps the orchestration of one evaluation (set-up, parameter upload, mode decision, begin / finish).  The
// other host-side pieces live in engine_rows / engine_linalg / engine_comm / engine_graph / engine_optim / abi .hip; shared
// declarations in engine_impl.h.
#include "engine_impl.h"

hipEvent_t hmogp_engine::new_event() {
  if (pool_used == pool.size()) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    pool.push_back(e);
  }
  return pool[pool_used++];
}

void hmogp_engine::collect_spans() {
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

hmogp_engine::~hmogp_engine() {
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
  if (h_cond) (void)hipHostFree(h_cond);
  if (ev_cond) (void)hipEventDestroy(ev_cond);
  if (st2_own) (void)hipStreamDestroy(st2_own);
  if (st3_own) (void)hipStreamDestroy(st3_own);
  if (st) (void)hipStreamDestroy(st);
}

void hmogp_engine::init(const hmogp_config* c) {
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

void hmogp_engine::set_task_data(int t, const double* X, const double* Y, long long N) {
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

void hmogp_engine::ensure_workspace(long long rows) {
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

void hmogp_engine::ensure_strict_workspace() {
  if (!strict) return;
  const long long MMs = (long long)M * M;
  D2.ensure(sizeof(double) * MMs * Q, true), dcond.ensure(sizeof(double) * HMOGP_MAXQ, true);
  if (!h_cond) HIP_TRY(hipHostMalloc((void**)&h_cond, sizeof(double) * HMOGP_MAXQ, hipHostMallocDefault));
  if (!ev_cond) HIP_TRY(hipEventCreateWithFlags(&ev_cond, hipEventDisableTiming));
  Dm.ensure(sizeof(double) * MMs * Q, true), Wq.ensure(sizeof(double) * MMs * Q, true), Lsy.ensure(sizeof(double) * MMs * Q, true);
  rdiag.ensure(sizeof(double) * (long long)M * Q, true), w3.ensure(sizeof(double) * (long long)M * Q, true);
  sVst = ((2LL * M + 1) * M + 1) & ~1LL;
  Vst.ensure(sizeof(double) * sVst * Q, true);
  if (ws_strict_rows >= ws_rows) return;
  trsmpart.ensure(sizeof(double) * 8 * ws_rows * Q);
  Ah.ensure(sizeof(double) * ws_rows * M * Q);
  vpg.ensure(sizeof(double) * ws_rows * Q, true), vcg.ensure(sizeof(double) * ws_rows * Q, true);
  ws_strict_rows = ws_rows;
}

void hmogp_engine::upload_params(const hmogp_params* p, bool enqueue) {
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

void hmogp_engine::decide_mode(const hmogp_params* p) {
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
  // (hmogp_params.eval_flags arrived with ABI v6: a caller that fills a v5-sized struct without zeroing it must get an error, not a
  //  silent strict step or a zero-filled g_L_u -- unknown bits are refused like those of hmogp_config.flags / quirks.  ADVICE r5)
  if (p && (p->eval_flags & ~(uint32_t)(HMOGP_EVAL_STRICT_QF | HMOGP_EVAL_NO_G_L)) != 0)
    throw EngineError{HMOGP_E_INVALID, "hmogp_params.eval_flags: unknown bits (HMOGP_EVAL_STRICT_QF | HMOGP_EVAL_NO_G_L are defined)"};
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

void hmogp_engine::begin(const hmogp_params* p, bool sync, bool will_exchange) {
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

bool hmogp_engine::small_failed() {
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

void hmogp_engine::fin_layout(const hmogp_outputs* out) {
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

void hmogp_engine::finish(hmogp_outputs* out) {
  finish_enqueue(out);
  finish_tail(out);
}

void hmogp_engine::finish_enqueue(hmogp_outputs* out) {
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
    if (strict && strict_two) {   // two-solve form: the bundle already holds dVE_dS = A^T diag(beta) A and dVE_dmu = A^T alpha (svmogp_inf.py:144-148)
      HIP_TRY(hipMemcpy2DAsync(G.p, sizeof(double) * MM, Hq(0), sizeof(double) * per_q, sizeof(double) * MM, Q, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpy2DAsync(Kr.p, sizeof(double) * M, Hq(0) + oR, sizeof(double) * per_q, sizeof(double) * M, Q, hipMemcpyDeviceToDevice, st));
    } else if (strict) {
      // [r6] one-solve form: the bundle holds H = X^T diag(beta) X and r = X^T alpha with X = K^ Luu^-T (both additive over rows and
      // ranks); dVE_dS = A^T diag(beta) A = Luu^-T H Luu^-1 and dVE_dmu = A^T alpha = Luu^-T r (svmogp_inf.py:144-148) follow by two
      // backward row-solves on M x M: [H ; r^T] Luu^-1 (M + 1 rows; its last row is (Luu^-T r)^T), transpose, once more, mirror.
      HIP_TRY(hipMemcpy2DAsync(Vst.p, sizeof(double) * sVst, Hq(0), sizeof(double) * per_q, sizeof(double) * MM, Q, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemcpy2DAsync(Vst.d() + MM, sizeof(double) * sVst, Hq(0) + oR, sizeof(double) * per_q, sizeof(double) * M, Q, hipMemcpyDeviceToDevice, st));
      potrs_rows_inplace(Vst.d(), sVst, Luu.d(), MM, M, M + 1, Q, st, Lsy.d(), nullptr, rdiag.d(), nullptr, 2, lsym_valid);
      HIP_TRY(hipMemcpy2DAsync(Kr.p, sizeof(double) * M, Vst.d() + MM, sizeof(double) * sVst, sizeof(double) * M, Q, hipMemcpyDeviceToDevice, st));
      launch_transpose_batched(Vst.d(), sVst, G.d(), MM, Q, M, st);
      potrs_rows_inplace(G.d(), MM, Luu.d(), MM, M, M, Q, st, Lsy.d(), nullptr, rdiag.d(), nullptr, 2, lsym_valid);
      launch_mirror_lower(G.d(), Q, M, MM, st);
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

void hmogp_engine::finish_tail(hmogp_outputs* out) {
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

void hmogp_engine::posterior_u(double* wv, double* winv) {
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
