// engine_linalg.hip -- host drivers of the batched dense linear algebra: GPy's jitchol ladder over potrf_step_kernel, the blocked
// triangular solves of the strict q(f) mode (dpotrs of svmogp_inf.py:214), launch-shape rules.
// Split out of engine.hip in round 6 (no behaviour change); declarations: engine_impl.h.
#include "engine_impl.h"

namespace hmogp_detail {

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

void jitchol_enqueue(const double* Kuu, double* Luu, int Q, int M, const double* diag_mean, int* rung_io, int* d_info,
                     double* d_jit, double* dscr, hipStream_t st, JitcholState& js, int part) {
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

bool potrs_rows_inplace(double* V, long long sV, const double* Luu, long long sL, int M, long long n, int Q, hipStream_t st,
                        double* Lsym, const double* Vsrc, double* rdiag, const TrsmRowStats* stats, int dirs, bool lsym_ready) {
  const bool fwd = (dirs & 1) != 0, bwd = (dirs & 2) != 0;
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
  // (the mirrored factor pays off from a few thousand rows on; the one-launch-per-block kernels also take the replicated
  //  M x M solves of the one-solve strict form -- 2 M + 1 and M + 1 rows -- where they replace 40 launches per direction by 8)
  const bool sym = Lsym != nullptr && n >= (rdiag ? 1024 : 4096) && sL == (long long)M * M;
  if (sym && !lsym_ready) {
    HIP_TRY(hipMemcpyAsync(Lsym, Luu, sizeof(double) * sL * Q, hipMemcpyDeviceToDevice, st));
    launch_mirror_lower(Lsym, Q, M, sL, st);       // Lsym[k][j] = Luu[j][k] above the diagonal
  }
  // [r6] ONE launch per 128-column block and direction (trsm_panel.hip): the block's long-K update and the substitution inside it
  // with the 128 x 128 tile in the accumulators throughout -- the right-hand sides are read once (from `Vsrc` by the forward
  // solve: no copy) and written once per solve, and the backward solve's epilogue forms the row statistics of A on the way out.
  // 16 launches at M = 1024 where round 5 needed 80; the round-5 path below stays for ragged M, short passes and A/B runs.
  if (sym && rdiag) {
    TrsmPanelArgs pa;
    pa.V = V, pa.sV = sV, pa.ldv = M, pa.Lsym = Lsym, pa.sL = sL, pa.ldl = M, pa.rdiag = rdiag, pa.sR = M, pa.n = n, pa.M = M, pa.Q = Q;
    pa.Vsrc = Vsrc;
    if (trsm_panel_eligible(pa)) {
      if (!lsym_ready) launch_rdiag(Luu, sL, M, Q, rdiag, M, st);
      auto with_stats = [&]() {
        if (!stats) return;
        pa.st_K = stats->K, pa.st_vec = stats->vec, pa.st_vecB = stats->vecB, pa.st_vecS = stats->vecS, pa.st_part = stats->part,
        pa.st_sPart = stats->sPart, pa.st_ld = stats->ld;
      };
      if (fwd) {
        if (!bwd) with_stats();
        for (int J0 = 0; J0 < M; J0 += 128) {                   // X Luu^T = V   (forward over the column blocks)
          pa.j0 = J0;
          launch_trsm_panel(0, pa, st);
        }
        pa.Vsrc = nullptr;
      }
      if (bwd) {
        with_stats();
        for (int J0 = M - 128; J0 >= 0; J0 -= 128) {            // A Luu = X     (backward)
          pa.j0 = J0;
          launch_trsm_panel(1, pa, st);
        }
      }
      return stats != nullptr;
    }
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
  if (Vsrc && fwd) {
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
  if (Vsrc && !fwd) {           // (a backward-only solve has no first-touch launches: its right-hand sides are copied)
    first_touch = false;
    for (int q = 0; q < Q; ++q)
      HIP_TRY(hipMemcpyAsync(V + q * sV, Vsrc + q * sV, sizeof(double) * n * M, hipMemcpyDeviceToDevice, st));
  }
  for (int J0 = 0; fwd && J0 < M; J0 += NB) {                 // X Luu^T = V   (forward over the columns)
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
  for (int J0 = ((M - 1) / NB) * NB; bwd && J0 >= 0; J0 -= NB) {   // A Luu = X     (backward over the columns)
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
  return false;
}

}  // namespace hmogp_detail
