// engine_optim.hip -- device-resident q(u): Adadelta, the natural-gradient step; predict_f; the raw-gradient debug export.
// Split out of engine.hip in round 6 (no behaviour change); declarations: engine_impl.h.
#include "engine_impl.h"

void hmogp_engine::qu_load(const double* m_u, const double* L_flat) {
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

void hmogp_engine::qu_read(double* m_u, double* L_flat) {
  if (!qu_resident) throw EngineError{HMOGP_E_STATE, "no resident q(u)"};
  HIP_TRY(hipSetDevice(device));
  if (m_u) HIP_TRY(hipMemcpyAsync(m_u, dmu.p, sizeof(double) * M * Q, hipMemcpyDeviceToHost, st));
  if (L_flat) HIP_TRY(hipMemcpyAsync(L_flat, dLflat.p, sizeof(double) * ((long long)M * (M + 1) / 2) * Q, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
}

void hmogp_engine::qu_adadelta(int phase, double rate, double m, double d, double omd, double o) {
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

void hmogp_engine::debug_raw(double* o_kmm, double* o_kmn, double* o_kdiag) {
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

void hmogp_engine::natgrad_core(double gamma, bool sync) {
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

void hmogp_engine::natgrad_step(double gamma, double* m_out, double* L_flat_out) {
  if (!m_out || !L_flat_out) throw EngineError{HMOGP_E_INVALID, "bad natural-gradient arguments"};
  natgrad_core(gamma);
  const long long Mtri = (long long)M * (M + 1) / 2;
  HIP_TRY(hipMemcpyAsync(m_out, ng_mq.p, sizeof(double) * M * Q, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(L_flat_out, ng_lflat.p, sizeof(double) * Mtri * Q, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
}

void hmogp_engine::qu_natgrad(double gamma) {
  if (!qu_resident) throw EngineError{HMOGP_E_STATE, "no resident q(u) (hmogp_qu_load)"};
  natgrad_core(gamma);
  const long long Mtri = (long long)M * (M + 1) / 2;
  HIP_TRY(hipMemcpyAsync(dmu.p, ng_mq.p, sizeof(double) * M * Q, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(dLflat.p, ng_lflat.p, sizeof(double) * Mtri * Q, hipMemcpyDeviceToDevice, st));
  // (no synchronisation: the next evaluation reads q(u) on streams ordered behind this one -- see upload_params)
  HIP_TRY(hipEventRecord(ev_qu, st));
}

void hmogp_engine::qu_natgrad_async(double gamma) {
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

int hmogp_engine::qu_natgrad_status() {
  if (ng_pending) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipEventSynchronize(ev_ng));
    ng_pending = false;
    ng_last_taken = true;
    for (int q = 0; q < Q; ++q) ng_last_taken = ng_last_taken && h_info2[q] == 0;
  }
  return ng_last_taken ? 1 : 0;
}

void hmogp_engine::predict_f(const double* Xnew, long long Nnew, double* m, double* v) {
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
      strict_forward(n, dX.d(), false, false);   // (in the form of the evaluation the prediction is taken from)
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
