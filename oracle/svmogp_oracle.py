"""TEST INFRASTRUCTURE ONLY -- NumPy CPU oracle for the svmogp_inf ELBO-and-gradient path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the product path (`hetmogp_amd/`) never does and fails loudly if its HIP library is missing.

Two restatements of the same mathematics, both pinned by `tests/golden/*.npz` (fixtures captured from
the reference's own Python, see `oracle/make_golden.py`; GPy 1.9.5 pieces are restated):

  * `inference_literal` + `assemble_literal`  ("baseline A") follow the reference step by step:
      hetmogp/util.py:145-200 (covariances), hetmogp/svmogp_inf.py:23-250 (q(f), VE, KL, raw gradients,
      including the materialised dL_dKmn) and hetmogp/svmogp.py:85-166 + util.py:228-231,248-255
      (parameter-gradient assembly).  `full_cov=True` also forms the N x N K_ff / S_fd the reference
      builds (svmogp_inf.py:49,202,219) -- used only to time "the reference CPU path".
  * `local_stats` + `finish`  ("baseline B") are the fused diag-only algebra the HIP engine implements
      (DESIGN.md section 3): per (task, latent) one N x M x M product forward, one weighted Gram backward,
      nothing N x N and no dL_dKmn.  `local_stats` is additive over row shards; `finish` is the
      replicated M x M post-processing -- the split the multi-GPU path all-reduces across.

Parameter container: dict with
  Z (M, Q*P)  m_u (M,Q)  L_flat (M(M+1)/2, Q) row-major tril  variance (Q,)  lengthscale (Q,)
  W (Q,Df)  kappa (Q,Df)  [W0 (Q,Df), kappa0 (Q,Df): construction-time copies, quirk Q3; default = live]
Problem container: dict with  specs [(name, kwargs)]*T,  f_index (Df,), d_index (Df,), Q, M, P.
"""
import numpy as np
import scipy.linalg
from scipy.linalg import lapack

from . import likelihoods_oracle as lo


# =============================================================================== helpers
def make_problem(specs, Q, M, P):
    """het_likelihood.py:24-44 (generate_metadata): function d belongs to task f_index[d], column d_index[d]."""
    f_index, d_index = [], []
    for t, (name, kw) in enumerate(specs):
        df = lo.dim_f(name, kw.get("K"))
        f_index += [t] * df
        d_index += list(range(df))
    return dict(specs=[(n, dict(k)) for n, k in specs], f_index=np.array(f_index), d_index=np.array(d_index),
                Q=Q, M=M, P=P, T=len(specs), Df=len(f_index))


def rbf_K(X, X2, variance, lengthscale):
    """GPy 1.9.5 RBF.K(X, X2) with X2 given: r2 = clip(|x|^2 + |x2|^2 - 2 x.x2, 0, inf) (util.py:161,178,197)."""
    X1sq = np.sum(np.square(X), 1)
    X2sq = np.sum(np.square(X2), 1)
    r2 = -2.0 * np.dot(X, X2.T) + (X1sq[:, None] + X2sq[None, :])
    r = np.sqrt(np.clip(r2, 0, np.inf)) / lengthscale
    return variance * np.exp(-0.5 * r ** 2)


def rbf_r2_scaled(X, X2, lengthscale, same=False):
    """Scaled squared distance; `same=True` is GPy's X2=None branch (diagonal forced to 0)."""
    X1sq = np.sum(np.square(X), 1)
    X2sq = np.sum(np.square(X2), 1)
    r2 = -2.0 * np.dot(X, X2.T) + (X1sq[:, None] + X2sq[None, :])
    if same:
        r2[np.diag_indices(X.shape[0])] = 0.0
    r = np.sqrt(np.clip(r2, 0, np.inf)) / lengthscale
    return r ** 2


def jitchol(A, maxtries=5, forced_rung=None):
    """GPy jitchol (util.py:198): plain dpotrf; on failure jitter = mean(diag)*1e-6*10^k, k = 0..4, added to
    the factorised matrix only.  Returns (L, rung) with rung = -1 for "no jitter".  `forced_rung` skips the
    search (used to compare CPU/GPU at equal rung)."""
    A = np.ascontiguousarray(A)
    if forced_rung is not None:
        if forced_rung < 0:
            L, info = lapack.dpotrf(A, lower=1)
            if info != 0:
                raise np.linalg.LinAlgError("forced rung -1 failed")
            return L, -1
        jit = np.diag(A).mean() * 1e-6 * 10.0 ** forced_rung
        return scipy.linalg.cholesky(A + np.eye(A.shape[0]) * jit, lower=True), forced_rung
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L, -1
    d = np.diag(A)
    if np.any(d <= 0.0):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    jit = d.mean() * 1e-6
    for k in range(maxtries):
        try:
            return scipy.linalg.cholesky(A + np.eye(A.shape[0]) * jit, lower=True), k
        except Exception:
            jit *= 10
    raise np.linalg.LinAlgError("not positive definite, even with jitter.")


def potri_sym(L):
    """GPy dpotri: (L L^T)^-1 from the lower factor, lower triangle mirrored up."""
    R, info = lapack.dpotri(np.asfortranarray(L), lower=1)
    R = np.array(R)
    iu = np.triu_indices(R.shape[0], 1)
    R[iu] = R.T[iu]
    return R, info


def flat_to_tril(flat_col, M):
    """GPy choleskies.flat_to_triang for one latent: row-major tril order."""
    L = np.zeros((M, M))
    L[np.tril_indices(M)] = flat_col
    return L


def tril_to_flat(A):
    return A[np.tril_indices(A.shape[0])]


def latent_covariances(prm, prob, forced_rungs=None):
    """util.py:181-200."""
    Q, M, P = prob["Q"], prob["M"], prob["P"]
    Kuu, Luu, Kuui, rungs = [], [], [], []
    for q in range(Q):
        Zq = prm["Z"][:, q * P:(q + 1) * P]
        K = rbf_K(Zq, Zq, prm["variance"][q], prm["lengthscale"][q])
        L, rung = jitchol(K, forced_rung=None if forced_rungs is None else forced_rungs[q])
        Ki, _ = potri_sym(L)
        Kuu.append(K), Luu.append(L), Kuui.append(Ki), rungs.append(rung)
    return np.stack(Kuu), np.stack(Luu), np.stack(Kuui), rungs


def _task_functions(prob, t):
    return [d for d in range(prob["Df"]) if prob["f_index"][d] == t]


def variational_expectations(prob, Y, mu_F, v_F, batch_scale):
    """het_likelihood.py:101-131 + svmogp_inf.py:73-78."""
    VE, VE_dm, VE_dv = [], [], []
    for t, (name, kw) in enumerate(prob["specs"]):
        ve, dm, dv = lo.var_exp_all(name, Y[t], mu_F[t], v_F[t], **kw)
        VE.append(ve * batch_scale[t]), VE_dm.append(dm * batch_scale[t]), VE_dv.append(dv * batch_scale[t])
    return VE, VE_dm, VE_dv


# ===================================================================== baseline A: literal
def q_f_literal(prm, prob, X, d, Kuu, Luu, Kuui, full_cov=False):
    """svmogp_inf.py:186-225 for one output function d.  Returns m_fd, v_fd, Afdu (Q,N,M), Kfdu (N,QM)."""
    Q, M, P = prob["Q"], prob["M"], prob["P"]
    N = X.shape[0]
    Kfdu = np.empty((N, Q * M))
    Kff_diag = np.zeros(N)
    for q in range(Q):
        Zq = prm["Z"][:, q * P:(q + 1) * P]
        Kfdu[:, q * M:(q + 1) * M] = prm["W"][q, d] * rbf_K(X, Zq, prm["variance"][q], prm["lengthscale"][q])
        Bdd = prm["W"][q, d] ** 2 + prm["kappa"][q, d]
        if full_cov:
            Kff_q = Bdd * rbf_K(X, X, prm["variance"][q], prm["lengthscale"][q])       # N x N (util.py:166-179)
            Kff_diag += np.diag(Kff_q)
        else:
            Kff_diag += Bdd * prm["variance"][q]          # diag of B_q[d,d] k_q(X,X): the only part read (:203)
    Afdu = np.empty((Q, N, M))
    m_fd = np.zeros(N)
    v_fd = Kff_diag.copy()
    for q in range(Q):
        L_q = flat_to_tril(prm["L_flat"][:, q], M)
        Kq = Kfdu[:, q * M:(q + 1) * M]
        R, _ = lapack.dpotrs(np.asfortranarray(Luu[q]), Kq.T, lower=1)          # Kuui Kuf
        Afdu[q] = R.T
        m_fd += Afdu[q] @ prm["m_u"][:, q]
        tmp = L_q.T @ R
        v_fd += np.sum(np.square(tmp), 0) - np.sum(R * Kq.T, 0)
        if full_cov:
            S_q = L_q @ L_q.T
            _ = (R.T @ S_q) @ R - Kq @ R                                          # S_fd term, N x N (:219)
    return m_fd, v_fd, Afdu, Kfdu


def kl_literal(prm, prob, Luu, Kuui):
    """svmogp_inf.py:227-250."""
    Q, M = prob["Q"], prob["M"]
    KL = 0.0
    for q in range(Q):
        L_q = flat_to_tril(prm["L_flat"][:, q], M)
        S_q = L_q @ L_q.T
        m = prm["m_u"][:, q]
        KL += 0.5 * np.sum(Kuui[q] * S_q) + 0.5 * m @ (Kuui[q] @ m) - 0.5 * M \
            + np.sum(np.log(np.abs(np.diag(Luu[q])))) - np.sum(np.log(np.abs(np.diag(L_q))))
    return KL


def inference_literal(prm, prob, X, Y, batch_scale=None, forced_rungs=None, full_cov=False):
    """svmogp_inf.py:23-109.  Returns dict(elbo, KL, m_fd[d], v_fd[d], grads{...}, VE_dm, VE_dv, rungs)."""
    Q, M, T, Df = prob["Q"], prob["M"], prob["T"], prob["Df"]
    f_index, d_index = prob["f_index"], prob["d_index"]
    batch_scale = [1.0] * T if batch_scale is None else list(batch_scale)
    Kuu, Luu, Kuui, rungs = latent_covariances(prm, prob, forced_rungs)
    qF = [q_f_literal(prm, prob, X[f_index[d]], d, Kuu, Luu, Kuui, full_cov) for d in range(Df)]
    mu_F = [np.stack([qF[d][0] for d in _task_functions(prob, t)], 1) for t in range(T)]
    v_F = [np.stack([qF[d][1] for d in _task_functions(prob, t)], 1) for t in range(T)]
    VE, VE_dm, VE_dv = variational_expectations(prob, Y, mu_F, v_F, batch_scale)
    KL = kl_literal(prm, prob, Luu, Kuui)
    elbo = sum(v.sum() for v in VE) - KL

    g = dict(dL_dmu_u=[], dL_dL_u=[], dL_dKmm=[], dL_dKmn=[], dL_dKdiag=[])
    for q in range(Q):                                                   # svmogp_inf.py:111-183
        L_q = flat_to_tril(prm["L_flat"][:, q], M)
        S_q = L_q @ L_q.T
        m = prm["m_u"][:, q:q + 1]
        Ki = Kuui[q]
        S_qi, _ = potri_sym(L_q)
        if np.any(np.isinf(S_qi)):
            raise ValueError("Sqi: Cholesky representation unstable")
        a = Ki @ m
        dKL_dmu = a
        dKL_dS = 0.5 * (Ki - S_qi)
        dKL_dK = 0.5 * Ki - 0.5 * Ki @ S_q @ Ki - 0.5 * (Ki @ (m @ m.T)) @ Ki.T
        dVE_dmu = np.zeros((M, 1))
        dVE_dS = np.zeros((M, M))
        dVE_dK = np.zeros((M, M))
        dKmn, dKdiag = [], []
        SKi2 = 2.0 * (S_q @ Ki - np.eye(M))
        for d in range(Df):
            t, j = f_index[d], d_index[d]
            A = qF[d][2][q]                                              # N x M
            gm, gv = VE_dm[t][:, j], VE_dv[t][:, j]
            dVE_dmu += (A.T @ gm)[:, None]
            Adv = A.T * gv[None, :]
            AdvA = Adv @ A
            dVE_dS += AdvA
            tmp_dv = AdvA @ S_q @ Ki
            dVE_dK += AdvA - tmp_dv - tmp_dv.T
            dVE_dK += -(A.T @ gm[:, None]) @ a.T
            dKmn.append(a @ gm[None, :] + SKi2.T @ Adv)
            dKdiag.append(gv)
        dVE_dK = 0.5 * (dVE_dK + dVE_dK.T)
        dL_dS = dVE_dS - dKL_dS
        g["dL_dmu_u"].append(dVE_dmu - dKL_dmu)
        g["dL_dL_u"].append(tril_to_flat(2.0 * dL_dS @ L_q)[:, None])
        g["dL_dKmm"].append(dVE_dK - dKL_dK)
        g["dL_dKmn"].append(dKmn)
        g["dL_dKdiag"].append(dKdiag)
    return dict(elbo=elbo, KL=KL, m_fd=[q[0] for q in qF], v_fd=[q[1] for q in qF], grads=g, VE_dm=VE_dm,
                VE_dv=VE_dv, rungs=rungs, Kuu=Kuu, Luu=Luu, Kuui=Kuui)


def assemble_literal(prm, prob, X, grads, stochastic=False, vem_step=True, z_fixed=False):
    """svmogp.py:85-166 + util.py:228-231,248-255 with GPy 1.9.5 RBF gradient formulas (restated, appendix A).
    Returns dict(g_m_u (M,Q), g_L_u (Mtri,Q), g_variance (Q,), g_lengthscale (Q,), g_W (Q,Df), g_kappa (Q,Df),
    g_Z (M,Q*P))."""
    Q, M, P, Df = prob["Q"], prob["M"], prob["P"], prob["Df"]
    f_index = prob["f_index"]
    W0 = prm.get("W0", prm["W"])
    kappa0 = prm.get("kappa0", prm["kappa"])
    e_gate = 0.0 if (stochastic and not vem_step) else 1.0               # q(u) groups: zero in M-steps
    m_gate = 0.0 if (stochastic and vem_step) else 1.0                   # hyper groups: zero in E-steps
    out = dict(g_m_u=np.zeros((M, Q)), g_L_u=np.zeros((M * (M + 1) // 2, Q)), g_variance=np.zeros(Q),
               g_lengthscale=np.zeros(Q), g_W=np.zeros((Q, Df)), g_kappa=np.zeros((Q, Df)),
               g_Z=np.zeros((M, Q * P)))
    for q in range(Q):
        var, ell = prm["variance"][q], prm["lengthscale"][q]
        Zq = prm["Z"][:, q * P:(q + 1) * P]
        out["g_m_u"][:, q] = e_gate * grads["dL_dmu_u"][q][:, 0]
        out["g_L_u"][:, q] = e_gate * grads["dL_dL_u"][q][:, 0]
        # K_uu part (X2=None branch)
        r2 = rbf_r2_scaled(Zq, Zq, ell, same=True)
        Kzz = var * np.exp(-0.5 * r2)
        dKmm = grads["dL_dKmm"][q]
        gvar = np.sum(Kzz * dKmm) / var
        gell = np.sum(dKmm * Kzz * r2) / ell
        gW = np.zeros(Df)
        gkap = np.zeros(Df)
        gZ = np.zeros((M, P))
        T2 = dKmm + dKmm.T
        # quirk Q10 (GPy 1.9.5 Stationary.gradients_X: `invdist = 1 / where(r != 0, r, inf)`): entries whose COMPUTED distance is
        # exactly 0 are dropped.  In exact arithmetic r = 0 means x = z and the term vanishes anyway; with un-centred inputs the
        # expanded form |x|^2 + |z|^2 - 2 x.z clips small positive distances to 0 and the dropped terms are visible (2e-5 of g_Z
        # for x in [1e4, 1e4 + 1], tests/golden/lad_c1_offset_rung1.npz)
        for p in range(P):
            gZ[:, p] += np.sum(-T2 * Kzz * (r2 != 0.0) * (Zq[:, p][:, None] - Zq[:, p][None, :]), 1) / ell ** 2
        for d in range(Df):
            Xt = X[f_index[d]]
            r2x = rbf_r2_scaled(Zq, Xt, ell)
            Kzx = var * np.exp(-0.5 * r2x)                                # M x N
            dK = grads["dL_dKmn"][q][d]
            sgv = np.sum(grads["dL_dKdiag"][q][d])
            gW[d] = prm["W"][q, d] * sgv + np.sum(dK * Kzx)               # util.py:230 (quirk Q4) + :252
            gkap[d] = sgv                                                 # util.py:231 (+0 from :250)
            gvar += W0[q, d] * np.sum(Kzx * dK) / var + (W0[q, d] ** 2 + kappa0[q, d]) * sgv
            gell += W0[q, d] * np.sum(dK * Kzx * r2x) / ell
            for p in range(P):
                gZ[:, p] += W0[q, d] * np.sum(-dK * Kzx * (r2x != 0.0) * (Zq[:, p][:, None] - Xt[:, p][None, :]), 1) / ell ** 2
        out["g_variance"][q] = m_gate * gvar
        out["g_lengthscale"][q] = m_gate * gell
        out["g_W"][q] = m_gate * gW
        out["g_kappa"][q] = m_gate * gkap
        if not z_fixed:
            out["g_Z"][:, q * P:(q + 1) * P] = m_gate * gZ
    return out


# ================================================================= baseline B: fused algebra
def stats_layout(prob):
    """Offsets (float64 words) of the additive statistic bundle (same layout as the HIP engine):
    global  [0] sum of scaled VE | [1] number of rows with v<0 | [2, 2+Df) sgv[d] = sum_n gv_nd
    per q   H (M*M) | r (M) | dZ (M*P) | sa | sl | swk (Df)          starting at  NG + q*per_q."""
    Q, M, P, Df = prob["Q"], prob["M"], prob["P"], prob["Df"]
    per_q = M * M + M + M * P + 2 + Df
    NG = 2 + Df
    return dict(NG=NG, per_q=per_q, size=NG + Q * per_q, H=0, r=M * M, dZ=M * M + M, sa=M * M + M + M * P,
                sl=M * M + M + M * P + 1, swk=M * M + M + M * P + 2, sgv=2)


def u_algebra(prm, prob, forced_rungs=None):
    """Replicated M x M quantities needed before the row pass: per q  Kuu, Luu, Kuui, L, S, a, C."""
    Q, M = prob["Q"], prob["M"]
    Kuu, Luu, Kuui, rungs = latent_covariances(prm, prob, forced_rungs)
    u = dict(Kuu=Kuu, Luu=Luu, Kuui=Kuui, rungs=rungs, L=[], S=[], a=[], C=[], D=[])
    for q in range(Q):
        L_q = flat_to_tril(prm["L_flat"][:, q], M)
        S_q = L_q @ L_q.T
        u["L"].append(L_q), u["S"].append(S_q)
        u["a"].append(Kuui[q] @ prm["m_u"][:, q])
        u["C"].append(Kuui[q] @ S_q @ Kuui[q] - Kuui[q])
        if prob.get("strict_qf"):      # the engine's HMOGP_CFG_STRICT_QF algebra (see local_stats)
            u["D"].append(S_q @ Kuui[q] - np.eye(M))                   # svmogp_inf.py:157-158 (tmp / 2)
    # [r6] which strict form (engine_impl.h: strict_two): the one-solve form while the condition estimate variance max diag(Kuu^-1)
    # of every latent is <= 1e6, the literal two-solve form beyond
    sq = prob.get("strict_qf")
    if sq:
        est = max(float(prm["variance"][q] * np.max(np.diag(Kuui[q]))) for q in range(Q))
        # (the engine's rule: the one-solve form up to the estimate 1e6, the two-solve form beyond; "two_solves" / "one_solve" force)
        u["strict_two"] = (sq == "two_solves") or (sq != "one_solve" and not (est <= 1e6))
    return u


def local_stats(prm, prob, u, X, Y, batch_scale=None):
    """Row pass over this shard's rows (additive over shards).  Returns the flat statistic bundle.

    prob["strict_qf"] = True restates the engine's HMOGP_CFG_STRICT_QF mode: q(f) and the row side of the gradients through
    triangular solves against Luu (the reference's dpotrs, svmogp_inf.py:214-218) -- since round 6 with only the FORWARD half of
    that solve on the n x M side (X = K^ Luu^-T; see the branch below), the bundle's H / r slots hold X^T diag(beta) X and X^T alpha
    and `finish` turns them into dVE_dS / dVE_dmu (:144-148); P~ = A (S Kuu^-1 - I) = X (Luu^-1 (S Kuu^-1 - I)) (:157-161).
    Beyond a condition estimate of 1e6 (and with prob["strict_qf"] = "two_solves"): the literal two-solve form (A = dpotrs, H = A^T
    diag(beta) A in the bundle), as in round 5; prob["strict_qf"] = "one_solve" forces the other one."""
    Q, M, P, T, Df = prob["Q"], prob["M"], prob["P"], prob["T"], prob["Df"]
    lay = stats_layout(prob)
    batch_scale = [1.0] * T if batch_scale is None else list(batch_scale)
    exact = prob.get("quirks", "reference") == "exact"      # not the reference: true gradients (see finish)
    W0 = prm["W"] if exact else prm.get("W0", prm["W"])
    stats = np.zeros(lay["size"])
    v_neg = False
    for t in range(T):
        ds = _task_functions(prob, t)
        Xt = X[t]
        N = Xt.shape[0]
        if N == 0:
            continue
        strict = bool(prob.get("strict_qf"))
        Khat, R2, Pt, p, c, pt, ct, Am, pg, cg = [], [], [], [], [], [], [], [], [], []
        for q in range(Q):
            Zq = prm["Z"][:, q * P:(q + 1) * P]
            ell = prm["lengthscale"][q]
            r2 = rbf_r2_scaled(Xt, Zq, ell)
            K = prm["variance"][q] * np.exp(-0.5 * r2)
            if strict and not u["strict_two"]:
                # [r6] the engine's strict mode (HMOGP_CFG_STRICT_QF) since round 6, while the condition estimate of K_uu is <= 1e6
                # (u_algebra: strict_two): ONE triangular solve on the n x M side.  With X = K^ Luu^-T
                # (forward substitution of dpotrs, svmogp_inf.py:214) the second half of the solve moves onto the M x M side:
                #   A m = X (Luu^-1 m),  A L_q = X (Luu^-1 L_q),  rowsum(A .* K^) = rowsum(X .* X)  (K^ = X Luu^T),
                #   A (S Kuu^-1 - I) = X (Luu^-1 (S Kuu^-1 - I)),  A^T diag(b) A = Luu^-T (X^T diag(b) X) Luu^-1 (finish)
                # -- the same quantities as the reference's forms, each through triangular SOLVES against Luu (never through the
                # explicit difference K^-1 S K^-1 - K^-1 whose cancellation costs the default path cond(K_uu) digits).
                Xm = scipy.linalg.solve_triangular(u["Luu"][q], K.T, lower=True).T
                PP = Xm @ scipy.linalg.solve_triangular(u["Luu"][q], u["D"][q], lower=True)
                Tm = Xm @ scipy.linalg.solve_triangular(u["Luu"][q], u["L"][q], lower=True)
                Am.append(Xm)
                p.append(Xm @ scipy.linalg.solve_triangular(u["Luu"][q], prm["m_u"][:, q], lower=True))
                c.append(np.sum(Tm * Tm, 1) - np.sum(Xm * Xm, 1))
                pg.append(K @ u["a"][q]), cg.append(np.sum(PP * K, 1))
            elif strict:
                V = scipy.linalg.solve_triangular(u["Luu"][q], K.T, lower=True)                  # two triangular SOLVES
                A = scipy.linalg.solve_triangular(u["Luu"][q], V, lower=True, trans="T").T       # (= dpotrs, :214)
                PP = A @ u["D"][q]
                Tm = A @ u["L"][q]
                Am.append(A)
                p.append(A @ prm["m_u"][:, q]), c.append(np.sum(Tm * Tm, 1) - np.sum(A * K, 1))
                pg.append(K @ u["a"][q]), cg.append(np.sum(PP * K, 1))
            else:
                PP = K @ u["C"][q]
                p.append(K @ u["a"][q]), c.append(np.sum(PP * K, 1))
            Khat.append(K), R2.append(r2), Pt.append(PP)
            pt.append((K * r2) @ u["a"][q]), ct.append(np.sum(PP * K * r2, 1))
        if not strict:
            pg, cg = p, c
        mu = np.zeros((N, len(ds)))
        vv = np.zeros((N, len(ds)))
        for j, d in enumerate(ds):
            for q in range(Q):
                w = prm["W"][q, d]
                mu[:, j] += w * p[q]
                vv[:, j] += (w * w + prm["kappa"][q, d]) * prm["variance"][q] + w * w * c[q]
        v_neg |= bool((vv < 0).any())
        name, kw = prob["specs"][t]
        ve, gm, gv = lo.var_exp_all(name, Y[t], mu, vv, exact=exact, **kw)
        ve, gm, gv = ve * batch_scale[t], gm * batch_scale[t], gv * batch_scale[t]
        stats[0] += ve.sum()
        stats[1] += float((vv < 0).sum())
        for j, d in enumerate(ds):
            stats[lay["sgv"] + d] += gv[:, j].sum()
        for q in range(Q):
            o = lay["NG"] + q * lay["per_q"]
            w = prm["W"][q, ds]
            alpha, beta = gm @ w, gv @ (w * w)
            alpha0, beta0 = gm @ W0[q, ds], gv @ (W0[q, ds] * w)
            K = Khat[q]
            Kg = Am[q] if strict else K
            stats[o + lay["H"]:o + lay["H"] + M * M] += ((Kg * beta[:, None]).T @ Kg).reshape(-1)
            stats[o + lay["r"]:o + lay["r"] + M] += Kg.T @ alpha
            E = (alpha0[:, None] * u["a"][q][None, :] + 2.0 * beta0[:, None] * Pt[q]) * K       # N x M
            Zq = prm["Z"][:, q * P:(q + 1) * P]
            for pp in range(P):
                stats[o + lay["dZ"] + pp:o + lay["dZ"] + M * P:P] += np.sum(
                    E * (R2[q] != 0.0) * (Xt[:, pp][:, None] - Zq[:, pp][None, :]), 0)     # (quirk Q10, see assemble_literal)
            stats[o + lay["sa"]] += alpha0 @ pg[q] + 2.0 * beta0 @ cg[q]
            stats[o + lay["sl"]] += alpha0 @ pt[q] + 2.0 * beta0 @ ct[q]
            for j, d in enumerate(ds):
                stats[o + lay["swk"] + d] += gm[:, j] @ pg[q] + 2.0 * prm["W"][q, d] * (gv[:, j] @ cg[q])
    return stats, v_neg


def finish(prm, prob, u, stats, stochastic=False, vem_step=True, z_fixed=False):
    """Replicated M x M post-processing of the (all-reduced) statistic bundle -> ELBO + parameter gradients."""
    Q, M, P, Df = prob["Q"], prob["M"], prob["P"], prob["Df"]
    lay = stats_layout(prob)
    # prob["quirks"] = "exact" (default "reference"): W0 := W (Q3), dW diag term 2 W variance sum(gv) (Q4), dkappa =
    # variance sum(gv) (Q5), and exact likelihood derivatives in local_stats (Q1, Q2): every gradient is then the true
    # gradient of the returned ELBO.  The reference mode is what the golden fixtures pin.
    exact = prob.get("quirks", "reference") == "exact"
    W0 = prm["W"] if exact else prm.get("W0", prm["W"])
    kappa0 = prm["kappa"] if exact else prm.get("kappa0", prm["kappa"])
    e_gate = 0.0 if (stochastic and not vem_step) else 1.0
    m_gate = 0.0 if (stochastic and vem_step) else 1.0
    out = dict(g_m_u=np.zeros((M, Q)), g_L_u=np.zeros((M * (M + 1) // 2, Q)), g_variance=np.zeros(Q),
               g_lengthscale=np.zeros(Q), g_W=np.zeros((Q, Df)), g_kappa=np.zeros((Q, Df)),
               g_Z=np.zeros((M, Q * P)), dL_dS=[])
    KL = 0.0
    sgv = stats[lay["sgv"]:lay["sgv"] + Df]
    for q in range(Q):
        o = lay["NG"] + q * lay["per_q"]
        H = stats[o:o + M * M].reshape(M, M)
        r = stats[o + lay["r"]:o + lay["r"] + M]
        dZs = stats[o + lay["dZ"]:o + lay["dZ"] + M * P].reshape(M, P)
        sa, sl = stats[o + lay["sa"]], stats[o + lay["sl"]]
        swk = stats[o + lay["swk"]:o + lay["swk"] + Df]
        Ki, S, L_q, a = u["Kuui"][q], u["S"][q], u["L"][q], u["a"][q]
        m = prm["m_u"][:, q]
        var, ell = prm["variance"][q], prm["lengthscale"][q]
        KL += 0.5 * np.sum(Ki * S) + 0.5 * m @ a - 0.5 * M + np.sum(np.log(np.abs(np.diag(u["Luu"][q])))) \
            - np.sum(np.log(np.abs(np.diag(L_q))))
        S_qi, _ = potri_sym(L_q)
        if np.any(np.isinf(S_qi)):
            raise ValueError("Sqi: Cholesky representation unstable")
        if prob.get("strict_qf") and not u["strict_two"]:   # the bundle holds X^T diag(beta) X, X^T alpha (X = K^ Luu^-T)
            Y1 = scipy.linalg.solve_triangular(u["Luu"][q], H, lower=True, trans="T")            # Luu^-T H
            G = scipy.linalg.solve_triangular(u["Luu"][q], Y1.T, lower=True, trans="T")          # Luu^-T H Luu^-1 (H symmetric)
            G = 0.5 * (G + G.T)
            Kr = scipy.linalg.solve_triangular(u["Luu"][q], r, lower=True, trans="T")
        elif prob.get("strict_qf"):      # the bundle already holds dVE_dS and dVE_dmu
            G, Kr = H, r
        else:
            G = Ki @ H @ Ki
            Kr = Ki @ r
        KiS = Ki @ S
        GSK = G @ KiS.T
        dVE_dK = G - GSK - GSK.T - np.outer(Kr, a)
        dVE_dK = 0.5 * (dVE_dK + dVE_dK.T)
        dKL_dK = 0.5 * Ki - 0.5 * KiS @ Ki - 0.5 * np.outer(a, a)
        dKmm = dVE_dK - dKL_dK
        dL_dS = G - 0.5 * (Ki - S_qi)
        out["dL_dS"].append(dL_dS)
        out["g_m_u"][:, q] = e_gate * (Kr - a)
        out["g_L_u"][:, q] = e_gate * tril_to_flat(2.0 * dL_dS @ L_q)
        Zq = prm["Z"][:, q * P:(q + 1) * P]
        r2 = rbf_r2_scaled(Zq, Zq, ell, same=True)
        Kzz = var * np.exp(-0.5 * r2)
        EK = dKmm * Kzz
        gvar = np.sum(EK) / var + sa / var + np.sum((W0[q] ** 2 + kappa0[q]) * sgv)
        gell = np.sum(EK * r2) / ell + sl / ell
        gZ = dZs / ell ** 2
        T2 = EK + EK.T
        for pp in range(P):
            gZ[:, pp] += np.sum(T2 * (r2 != 0.0) * (Zq[:, pp][None, :] - Zq[:, pp][:, None]), 1) / ell ** 2   # (quirk Q10)
        out["g_variance"][q] = m_gate * gvar
        out["g_lengthscale"][q] = m_gate * gell
        if exact:
            out["g_W"][q] = m_gate * (2.0 * prm["W"][q] * var * sgv + swk)
            out["g_kappa"][q] = m_gate * var * sgv
        else:
            out["g_W"][q] = m_gate * (prm["W"][q] * sgv + swk)          # util.py:230 (quirk Q4) + :252
            out["g_kappa"][q] = m_gate * sgv                             # util.py:231 (quirk Q5)
        if not z_fixed:
            out["g_Z"][:, q * P:(q + 1) * P] = m_gate * gZ
    out["KL"] = KL
    out["elbo"] = stats[0] - KL
    return out


def elbo_grad_fused(prm, prob, X, Y, batch_scale=None, forced_rungs=None, **gates):
    u = u_algebra(prm, prob, forced_rungs)
    stats, v_neg = local_stats(prm, prob, u, X, Y, batch_scale)
    out = finish(prm, prob, u, stats, **gates)
    out["rungs"], out["v_negative"] = u["rungs"], v_neg
    return out


def elbo_grad_literal(prm, prob, X, Y, batch_scale=None, forced_rungs=None, full_cov=False, **gates):
    inf = inference_literal(prm, prob, X, Y, batch_scale, forced_rungs, full_cov)
    out = assemble_literal(prm, prob, X, inf["grads"], **gates)
    out["elbo"], out["KL"], out["rungs"] = inf["elbo"], inf["KL"], inf["rungs"]
    return out


# ===================================================================== prediction
def gpy_posterior(mean, cov, K, forced_rung=None):
    """GPy 1.9.5 `Posterior(mean, cov, K)` read through its lazy properties: K_chol = jitchol(K);
    woodbury_vector = dpotrs(K_chol, mean) = K^-1 mean; woodbury_inv = K^-1 (K - cov) K^-1 by two dpotrs.  Returns
    (woodbury_vector (N,1), woodbury_inv (N,N), rung)."""
    Kc, rung = jitchol(K, forced_rung=forced_rung)
    Kc = np.asfortranarray(Kc)
    wv, _ = lapack.dpotrs(Kc, np.asarray(mean, float).reshape(-1, 1), lower=1)
    tmp, _ = lapack.dpotrs(Kc, K - cov, lower=1)
    wi, _ = lapack.dpotrs(Kc, tmp.T, lower=1)
    return wv, wi, rung


def q_f_full_literal(prm, prob, X, d, Luu):
    """What inference() hands `Posterior` for output function d (svmogp_inf.py:43-51 with calculate_q_f :186-225):
    mean m_fd (N,), the FULL covariance S_fd (N,N) (:219) and K_ff = sum_q B_q[d,d] k_q(X,X) (util.py:166-179)."""
    Q, M, P = prob["Q"], prob["M"], prob["P"]
    N = X.shape[0]
    Kff = np.zeros((N, N))
    m_fd = np.zeros(N)
    S_fd = np.zeros((N, N))
    for q in range(Q):
        Zq = prm["Z"][:, q * P:(q + 1) * P]
        Kq = prm["W"][q, d] * rbf_K(X, Zq, prm["variance"][q], prm["lengthscale"][q])
        Kff += (prm["W"][q, d] ** 2 + prm["kappa"][q, d]) * rbf_K(X, X, prm["variance"][q], prm["lengthscale"][q])
        L_q = flat_to_tril(prm["L_flat"][:, q], M)
        R, _ = lapack.dpotrs(np.asfortranarray(Luu[q]), Kq.T, lower=1)
        m_fd += R.T @ prm["m_u"][:, q]
        S_fd += (R.T @ (L_q @ L_q.T)) @ R - Kq @ R
    return m_fd, S_fd + Kff, Kff


def _predict_from_posterior(prm, prob, Xbase, Xnew, d, wv, wi):
    """The common tail of _raw_predict_f / predictive_new / _raw_predict_stochastic (svmogp.py:267-278)."""
    Kx = np.zeros((Xbase.shape[0], Xnew.shape[0]))
    Kxx = np.zeros(Xnew.shape[0])
    for q in range(prob["Q"]):
        Bdd = prm["W"][q, d] ** 2 + prm["kappa"][q, d]
        Kx += Bdd * rbf_K(Xbase, Xnew, prm["variance"][q], prm["lengthscale"][q])
        Kxx += Bdd * np.diag(rbf_K(Xnew, Xnew, prm["variance"][q], prm["lengthscale"][q]))
    mu = Kx.T @ wv
    var = (Kxx - np.sum((wi @ Kx) * Kx, 0))[:, None]
    return mu, np.abs(var)


def raw_predict_f_literal(prm, prob, X, Xnew, d, forced_rungs=None):
    """SVMOGP._raw_predict_f (svmogp.py:255-278): q(f_d) at the TRAINING inputs of d's task is turned into a GPy Posterior
    (an N x N factorisation of K_ff) and regressed onto Xnew.  This is the route the reference's `predictive` and
    `negative_log_predictive` take (:333-370); it is NOT the same estimator as `predictive_new` / calculate_q_f at Xnew."""
    _, Luu, _, _ = latent_covariances(prm, prob, forced_rungs)
    Xt = X[prob["f_index"][d]]
    m, S, Kff = q_f_full_literal(prm, prob, Xt, d, Luu)
    wv, wi, _ = gpy_posterior(m, S, Kff)
    return _predict_from_posterior(prm, prob, Xt, Xnew, d, wv, wi)


def predictive_new_literal(prm, prob, Xnew, d, forced_rungs=None):
    """SVMOGP.predictive_new (svmogp.py:280-306): the Posterior is built AT Xnew, so up to the N_new x N_new solve the
    result is (m_fd(Xnew), |v_fd(Xnew)|) of calculate_q_f."""
    _, Luu, _, _ = latent_covariances(prm, prob, forced_rungs)
    m, S, Kff = q_f_full_literal(prm, prob, Xnew, d, Luu)
    wv, wi, _ = gpy_posterior(m, S, Kff)
    return _predict_from_posterior(prm, prob, Xnew, Xnew, d, wv, wi)


def raw_predict_u_literal(prm, prob, Xnew, q, block0=True, forced_rungs=None):
    """SVMOGP._raw_predict (svmogp.py:219-253), diagonal variance: the latent u_q at Xnew from posteriors[q] =
    Posterior(mean=m_q, cov=S_q, K=Kuu_q) (svmogp_inf.py:181).  block0=True reproduces `kern.K(self.Z, Xnew)` under GPy's
    input slicing: the kernel sees the first P columns of the M x (Q P) inducing array -- latent 0's block for every q
    (identical to block q while Z is the tiled initialisation, svmogp.py:52); block0=False uses block q."""
    P, M = prob["P"], prob["M"]
    Kuu, _, _, _ = latent_covariances(prm, prob, forced_rungs)
    L_q = flat_to_tril(prm["L_flat"][:, q], M)
    wv, wi, _ = gpy_posterior(prm["m_u"][:, q], L_q @ L_q.T, Kuu[q],
                              forced_rung=None if forced_rungs is None else forced_rungs[q])
    b = 0 if block0 else q
    Kx = rbf_K(prm["Z"][:, b * P:(b + 1) * P], Xnew, prm["variance"][q], prm["lengthscale"][q])
    mu = Kx.T @ wv
    var = (prm["variance"][q] - np.sum((wi @ Kx) * Kx, 0))[:, None]
    return mu, np.abs(var)


def predictive_literal(prm, prob, X, Xpred, trained=True):
    """SVMOGP.predictive (svmogp.py:333-351): q(f) through `_raw_predict_f`, then `<likelihood>.predictive` per task
    (het_likelihood.py:133-148).  trained=True: the model's likelihood instances have run var_exp, so Gamma / Beta
    read their cached 10-point rule (quirk Q7)."""
    m_out, v_out = [], []
    for t, (name, kw) in enumerate(prob["specs"]):
        ds = _task_functions(prob, t)
        mv = [raw_predict_f_literal(prm, prob, X, Xpred[t], d) for d in ds]
        m = np.hstack([a for a, _ in mv])
        v = np.hstack([b for _, b in mv])
        gh = 10 if (trained and name in ("Gamma", "Beta")) else None
        mp, vp = lo.predictive(name, m, v, gh_T=gh, **kw)
        m_out.append(mp), v_out.append(vp)
    return m_out, v_out


# ----------------------------------------------------------------------------- loaders
def load_case(npz):
    """Rebuild (prm, prob, X, Y, batch_scale) from an `inf_*.npz` / `model_*.npz` golden fixture."""
    import json
    specs = [(n, k) for n, k in json.loads(str(npz["spec"]))]
    Q, M, P, T = int(npz["Q"]), int(npz["M"]), int(npz["P"]), int(npz["T"])
    prob = make_problem(specs, Q, M, P)
    prm = dict(Z=npz["Z"], m_u=npz["m_u"], L_flat=npz["L_flat"], variance=npz["variance"],
               lengthscale=npz["lengthscale"], W=npz["W"], kappa=npz["kappa"])
    if "W0" in npz.files:
        prm["W0"] = npz["W0"]
    key = "Xbatch_%d" if "Xbatch_0" in npz.files else "X_%d"
    X = [npz[key % t] for t in range(T)]
    Y = [npz[key.replace("X", "Y") % t] for t in range(T)]
    return prm, prob, X, Y, list(npz["batch_scale"])
