// engine_rows.hip -- the replicated M x M chain in front of the rows (u_algebra) and the row pass (K^ -> P~ -> q(f) -> quadrature -> Gram / column statistics), default and strict q(f) forms.
// Split out of engine.hip in round 6 (no behaviour change); declarations: engine_impl.h.
#include "engine_impl.h"

void hmogp_engine::u_algebra_small() {
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

void hmogp_engine::u_algebra() {
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
  bool cond_pending = false;     // set by tail(): an early condition estimate is on its way (strict mode, new factor)
  if (!kuu_hit) lsym_valid = false;             // (a new factor: its mirrored image / pivot reciprocals are rebuilt by the first strict use)
  auto tail = [&](bool first) {
    if (!kuu_hit && strict) {
      // strict mode: K_uu^-1 = dpotrs(Luu, I) by the blocked substitution, lower triangle mirrored like GPy's dpotri wrapper
      // (util.py:199).  The merge-based triangular inverse below is ~100x further from LAPACK's dpotri where it matters here
      // (|K_uu^-1 K_uu - I| 8.6e-8 against 3e-10 at cond 1e7) -- invisible at cond <= 1e5, 2e-8 of g_W / g_Z at 1e7.
      if (first) HIP_TRY(hipStreamWaitEvent(st, ev_zero, 0));
      strict_factor_images();                     // [r6] the mirrored factor + its pivots' reciprocals: once per factorisation
      launch_identity(Kuui.d(), Q, M, st);
      potrs_rows_inplace(Kuui.d(), MM, Luu.d(), MM, M, M, Q, st, Lsy.d(), nullptr, rdiag.d(), nullptr, 3, true);
      launch_mirror_lower(Kuui.d(), Q, M, MM, st);
      launch_cond_probe(Kuui.d(), dvar.d(), Q, M, dcond.d(), st);      // which strict form: see strict_two (engine_impl.h)
      HIP_TRY(hipMemcpyAsync(h_cond, dcond.p, sizeof(double) * Q, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(ev_cond, st));
      cond_pending = true;
    } else if (!kuu_hit) {
      if (first) HIP_TRY(hipStreamWaitEvent(st, ev_zero, 0));     // (tmpA zeroed on the third stream, above)
      launch_trtri_batched(Luu.d(), tmpA.d(), tmpB.d(), Q, M, st, first);
      launch_ltl_batched(tmpA.d(), Kuui.d(), Q, M, st);           // K_uu^-1          (util.py:199)
    }
    launch_gemv_batched(Kuui.d(), dmu.d(), a.d(), Q, M, 1, Q, st);  // a = K_uu^-1 m
    HIP_TRY(hipStreamWaitEvent(st, ev_S, 0));
    mm(Kuui.d(), false, S.d(), true, KiS.d());
    if (strict) {
      // [r6] one-solve form: the right factors of A m, A L_q and A (S K_uu^-1 - I) (svmogp_inf.py:216-217, :157-159) with the
      // BACKWARD half of dpotrs already inside them -- one forward row-solve of 2 M + 1 rows on the M x M side instead of a second
      // pass of substitutions over the n x M side:  Wq = Luu^-1 L_q,  Dm = Luu^-1 (S K_uu^-1 - I),  w3 = Luu^-1 m
      if (!lsym_valid) strict_factor_images();   // (cached K_uu chain whose factor was last used by a default-mode evaluation)
      launch_strict_stack(L.d(), KiS.d(), dmu.d(), Vst.d(), sVst, Q, M, st);
      potrs_rows_inplace(Vst.d(), sVst, Luu.d(), MM, M, 2LL * M + 1, Q, st, Lsy.d(), nullptr, rdiag.d(), nullptr, 1, true);
      launch_strict_unstack(Vst.d(), sVst, Wq.d(), Dm.d(), w3.d(), Q, M, st);
      launch_strict_d(KiS.d(), D2.d(), Q, M, st);                    // S K_uu^-1 - I: the two-solve form's right factor (:157-158)
    }
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
      lsym_valid = false;
      tail(false);
    }
    if (cache_kuu) kuu_key.swap(key), kuu_rung = rung, kuu_key_valid = true;
  }
  if (strict && cond_pending) {     // (the device is still busy with the rest of the chain: this wait costs no device time)
    HIP_TRY(hipEventSynchronize(ev_cond));
    cond_two = false;
    for (int q = 0; q < Q; ++q) cond_two = cond_two || !(h_cond[q] <= 1e6);
  }
  if (strict) {
    static const int force = [] {   // HMOGP_STRICT_FORM=1 | 2: force the one-solve / two-solve form (A/B runs)
      const char* e = getenv("HMOGP_STRICT_FORM");
      return e ? atoi(e) : 0;
    }();
    // The one-solve form up to the condition estimate 1e6, the two-solve form beyond.  P~ = A (S Kuu^-1 - I) -- needed by the K_uf-side
    // gradients only -- is the one product whose one-solve form X (Luu^-1 (S Kuu^-1 - I)) drifts from the two-solve one as K_uu
    // degrades: form against form at M = 1024 (tools/strict_ab.py ... forms) the worst element-wise excess over the 1e-5 criterion is
    // 5e-4 at the estimate 2.5e3, 0.012 at 1.1e5, 0.30 at 5.5e5 (jitter rung 0) -- but against the REFERENCE'S OPERATIONS (the literal
    // LAPACK restatement, tools/forms_vs_literal.py) both forms sit at the SAME distance there: 3.87 vs 3.86 x the criterion in g_Z at
    // M = 1024, 3.17 vs 3.16 at M = 512 (two valid K_uu^-1 of a cond-1e7 matrix differ by 1e-9: the K_uu-side terms amplify that,
    // DESIGN 6a), and both pass the reference-run fixtures at M = 128 (rung 0, rung 1).  Beyond 1e6 -- the notebook's own
    // hyper-parameters, cond 1e12 -- the one-solve g_Z leaves 50 x the reference's own sensitivity (1.6e-2 against 9e-4): two solves.
    strict_two = cond_two;
    if (force == 1) strict_two = false;
    if (force == 2) strict_two = true;
  }
}

void hmogp_engine::plan_pools() {
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

void hmogp_engine::kuf_pool(const std::vector<Seg>& pl, hipStream_t stream, size_t seg_begin, size_t seg_end) {
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

void hmogp_engine::stage_pool_inputs(const std::vector<Seg>& pl, hipStream_t stream) {
  if (pl.size() <= 1) return;
  std::vector<long long> key;
  for (auto& sg : pl) key.push_back(sg.t), key.push_back(sg.r0), key.push_back(sg.n), key.push_back(sg.off);
  if (pools.size() == 1 && key == staged_key) return;
  staged_key = pools.size() == 1 ? key : std::vector<long long>();
  for (auto& sg : pl)
    HIP_TRY(hipMemcpyAsync(Xws.d() + sg.off * P, tasks[sg.t].X.d() + sg.r0 * P, sizeof(double) * sg.n * P,
                           hipMemcpyDeviceToDevice, stream));
}

void hmogp_engine::strict_factor_images() {
  const long long MM = (long long)M * M;
  HIP_TRY(hipMemcpyAsync(Lsy.p, Luu.p, sizeof(double) * MM * Q, hipMemcpyDeviceToDevice, st));
  launch_mirror_lower(Lsy.d(), Q, M, MM, st);       // Lsy[k][j] = Luu[j][k] above the diagonal
  launch_rdiag(Luu.d(), MM, M, Q, rdiag.d(), M, st);
  lsym_valid = true;
}

void hmogp_engine::strict_forward(long long n, const double* X, bool grads, bool hyper) {
  const long long MM = (long long)M * M, ldn = ws_rows, sK = ldn * M;
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
  // [r6] ONE-SOLVE FORM.  The reference forms R = dpotrs(Luu, K^T) (svmogp_inf.py:214) -- a forward and a backward substitution over
  // all n rows -- and then R^T m, dtrmm(L_q^T, R), sum(R * K^T), (:216-218) and R^T-sided products for the gradients (:144-161).  Here
  // only the FORWARD half touches the n x M side: X = K^ Luu^-T (into `Ah`); the backward half sits inside the M x M right factors
  // Wq / Dm / w3 (u_algebra) and inside the two M x M solves that turn X^T diag(beta) X, X^T alpha into dVE_dS, dVE_dmu (finish):
  //   A m = X w3,  A L_q = X Wq,  rowsum(A .* K^) = rowsum(X .* X)  (K^ = X Luu^T),  A (S Kuu^-1 - I) = X Dm.
  // Same quantities, every one through triangular solves against Luu (never through K^-1 S K^-1 - K^-1, whose cancellation is
  // what costs the default path cond(K_uu) digits); n M^2 flops and two passes over the n x M matrix less than two solves.
  // Against the reference's own runs in the ladder regime the two forms are equally close (oracle prototype).
  // The solve's epilogue leaves p = X w3 and rowsum(X .* X) where the one-launch-per-block kernels take it (trsm_panel.hip).
  TrsmRowStats ts;
  ts.part = trsmpart.d(), ts.sPart = 8 * ldn, ts.ld = ldn;
  if (strict_two) ts.K = Kh.d(), ts.vec = dmu.d(), ts.vecB = 1, ts.vecS = Q;        // two-solve form: p = A m, rowsum(A .* K^)
  else ts.K = nullptr, ts.vec = w3.d(), ts.vecB = M, ts.vecS = 1;                    // one-solve form: p = X w3, rowsum(X .* X)
  static const bool ts_env = [] {   // HMOGP_TRSM_STATS=0: the statistics by strict_rowstats_kernel (A/B runs)
    const char* e = getenv("HMOGP_TRSM_STATS");
    return !(e && e[0] == '0');
  }();
  bool stats_fused = false;
  {
    Scope sc(this, CAT_TRSM, (strict_two ? 2 : 1) * ((M + 127) / 128));
    stats_fused = potrs_rows_inplace(Ah.d(), sK, Luu.d(), MM, M, n, Q, st, Lsy.d(), Kh.d(), rdiag.d(), ts_env ? &ts : nullptr,
                                     strict_two ? 3 : 1, lsym_valid);
  }
  const double* Bt = strict_two ? L.d() : Wq.d();       // right factor of T:  A L_q  |  X (Luu^-1 L_q)      (both lower triangular)
  const double* Bp = strict_two ? D2.d() : Dm.d();      // right factor of P~: A (S Kuu^-1 - I)  |  X (Luu^-1 (S Kuu^-1 - I))
  // T = A L_q = X Wq = dtrmm(L_q^T, R)^T (:217) is only ever consumed as rowsum(T .* T) (:218): where the specialised fold kernel takes
  // the product, its epilogue forms that sum from the accumulators and T is neither written nor read back (2 x 19.7 GB at H)
  bool t2_fused = false;
  {
    Scope sc(this, CAT_FWD, 2);
    GemmArgs g;
    g.A = Ah.d(), g.lda = M, g.a_kmajor = 0, g.sA = sK;
    g.B = Bt, g.ldb = M, g.b_kmajor = 1, g.sB = MM, g.b_tri = +1;
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
      rows_gemm(Ah.d(), Bt, 1, +1, Pt.d());
    }
  }
  StrictRows sr;
  sr.M = M, sr.Q = Q, sr.P = P, sr.ldz = Q * P, sr.n = n, sr.ldn = ldn, sr.sK = sK, sr.sZ = P;
  sr.Kh = Kh.d(), sr.Ah = Ah.d(), sr.Tt = Pt.d(), sr.Pt = Pt.d(), sr.mu = dmu.d(), sr.w3 = strict_two ? nullptr : w3.d(), sr.a = a.d();
  sr.X = X, sr.Z = dZ.d(), sr.ell = dell.d();
  sr.p = vp.d(), sr.c = vc.d(), sr.pg = vpg.d(), sr.cg = vcg.d(), sr.pt = nullptr, sr.ct = nullptr;
  sr.phase = 0;
  sr.t2 = t2_fused ? vct.d() : nullptr;           // (vct: free until phase 1 writes the r2-weighted twin into it)
  {
    Scope sc(this, CAT_STRICT_STATS, 1);
    if (stats_fused && t2_fused)                  // p = A m, c = rowsum(T^2) - rowsum(A .* K^)   (:216, :218)
      launch_trsm_stats_combine(trsmpart.d(), ts.sPart, ldn, 4, n, Q, vct.d(), vp.d(), vc.d(), ldn, st);
    else
      launch_strict_rowstats(sr, st);
  }
  sr.t2 = nullptr;
  if (!grads) return;
  // P~ = A (S Kuu^-1 - I) (:157-161).  [r6] Where the specialised forward kernel takes the product its fused epilogue forms the two
  // row statistics the gradient code reduces P~ to -- pg = K^ a and cg = rowsum(P~ .* K^) -- from the accumulators and a K^ tile
  // read through `fs_k` (K^ is not this product's A operand), and the r2-weighted statistic of the lengthscale gradient comes
  // from the column statistics (colstats<STRICT, SL>: GPy's r2 form): no pass of strict_rowstats_kernel over K^ and P~
  // (39 GB, 8.7 ms at the headline size).  `hyper` then only says that colstats carries the weight.
  static const bool p1_env = [] {   // HMOGP_STRICT_P1=0: phase 1 by strict_rowstats_kernel as in round 5 (pg / cg only; A/B runs)
    const char* e = getenv("HMOGP_STRICT_P1");
    return !(e && e[0] == '0');
  }();
  {
    GemmArgs g;
    g.A = Ah.d(), g.lda = M, g.a_kmajor = 0, g.sA = sK;
    g.B = Bp, g.ldb = M, g.b_kmajor = 1, g.sB = MM;
    g.C = Pt.d(), g.ldc = M, g.sC = sK;
    g.M = (int)n, g.N = M, g.K = M;
    g.nbatch = Q;
    g.role = 1;
    const int tiles = (M + 127) / 128;
    g.fs_part = fwdpart.d(), g.fs_sPart = 4LL * FWD_PARTS * tiles * ldn, g.fs_a = a.d(), g.fs_sA = M, g.fs_k = Kh.d();
    if (p1_env && gemm_rowpass_would_take(g)) {
      {
        Scope sc(this, CAT_FWD, 1);
        launch_gemm_rowpass_or_general(g, st);
      }
      Scope sc(this, CAT_STRICT_STATS, 1);
      launch_combine_parts(fwdpart.d(), 4 * tiles, n, vpg.d(), vcg.d(), nullptr, nullptr, st, Q, g.fs_sPart, ldn);
      return;
    }
  }
  {
    Scope sc(this, CAT_FWD, 1);
    rows_gemm(Ah.d(), Bp, 1, 0, Pt.d());
  }
  Scope sc(this, CAT_STRICT_STATS, 1);
  sr.phase = 1;
  sr.pt = sr.ct = nullptr;                        // (the r2-weighted twins: column statistics now)
  launch_strict_rowstats(sr, st);                 // K^ a, rowsum(P~ .* K^)
}

void hmogp_engine::row_pass() {
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
    const bool col_sl = want_hyper && !small_rows && col_sl_env;     // ([r6] strict q(f) too: colstats<STRICT> weights with GPy's r2 form)
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
