#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- golden-vector generator (runs ONLY where /root/reference exists).

Imports the reference's OWN Python (`/root/reference/hetmogp/*.py`, `/root/reference/likelihoods/*.py`)
over `oracle/gpy_standin.py` and records input/output vectors as small `.npz` fixtures under
`tests/golden/`.  The fixtures are data (inputs + expected outputs); no reference source is
copied.  Re-run:  python oracle/make_golden.py

Fixture families:
  lik_<name>.npz    G1  per-likelihood var_exp / var_exp_derivatives          (likelihoods/*.py)
  cov_<case>.npz    G2  Kuu, Luu (jitchol), Kuui, ladder rung                  (util.py:181-200)
  inf_<case>.npz    G3/G4/G5  q(f_d), KL, ELBO and the raw gradient dict        (svmogp_inf.py:23-250)
  pred_<name>.npz   f2  per-likelihood predictive mean / variance                 (likelihoods/*.py `predictive`)
  model_<case>.npz  G6  assembled parameter gradients from the reference's own
                        SVMOGP.parameters_changed (svmogp.py:85-166) run over the stand-in's
                        RESTATED GPy RBF gradient formulas ("GPy-unpinned").
  ref_<case>.npz    N1  REAL-SIZE runs of the reference's own SVMOGP.parameters_changed + SVMOGPInf.inference (C1 exactly:
                        N_t = 1000, M = 50, Q = 2; multi-tile M = 128 / 144 / 160): ELBO, KL, q(f_d), dL_dmu_u, dL_dL_u, dL_dKmm
                        and the assembled parameter gradients; the Q*Df dense M x N `dL_dKmn` blocks are NOT stored (size).
  lad_<case>.npz    J1  the reference's own SVMOGP.parameters_changed in the regime where GPy's jitchol decides the numbers
                        (notebook hyper-parameters at C1's shape: cond 1e12, rung -1; M = 128 at 4 spacings: rung 0; un-centred
                        inputs: rung 1), each run twice (inputs, inputs moved by one ulp): `sens_*` = the reference's own
                        rounding-level sensitivity.
  mpred_<case>.npz  f2  model-level prediction through the reference's own SVMOGP methods (svmogp.py:219-351):
                        predictive_new, _raw_predict_f, _raw_predict_stochastic, _raw_predict, predictive, at tiny N
                        (the _raw_predict_f route factorises the N x N K_ff of the TRAINING inputs).  Relies on the
                        stand-in's restated GPy `Posterior` (lazy woodbury_vector / woodbury_inv) and kernel input
                        slicing ("GPy-unpinned"); the inputs are spaced so that no K_ff needs jitter (rungs recorded).
  svi_traj_<case>.npz f1  13 consecutive iterations of the reference's own SVI driver -- util.vem_algorithm(stochastic=True)
                        (util.py:316-329) over SVMOGP.stochastic_grad / new_batch / callback (svmogp.py:168-217) and
                        draw_mini_slices (util.py:52-72); paramz (`_grads`, `optimizer_array`, Logexp, regexp fix) and climin's
                        Adadelta are the stand-in's restatements ("paramz/climin-unpinned"); slice bounds, E/M gating, ELBO,
                        optimiser vector and gradient per iteration.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("HETMOGP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    from oracle import gpy_standin
    gpy_standin.install()
    import hetmogp.svmogp_inf as inf
    import hetmogp.util as util
    import hetmogp.het_likelihood as hl
    import hetmogp.svmogp as svmogp
    from likelihoods import gaussian, bernoulli, hetgaussian, categorical, poisson, gamma, beta, exponential
    liks = dict(Gaussian=gaussian.Gaussian, Bernoulli=bernoulli.Bernoulli, HetGaussian=hetgaussian.HetGaussian,
                Categorical=categorical.Categorical, Poisson=poisson.Poisson, Gamma=gamma.Gamma,
                Beta=beta.Beta, Exponential=exponential.Exponential)
    return gpy_standin, inf, util, hl, svmogp, liks


DIM_F = dict(Gaussian=1, Bernoulli=1, HetGaussian=2, Poisson=1, Gamma=2, Beta=2, Exponential=1)


def dim_f(spec):
    name, kw = spec
    return kw["K"] - 1 if name == "Categorical" else DIM_F[name]


def make_lik(liks, spec):
    name, kw = spec
    return liks[name](**kw)


def sample_y(rng, spec, n):
    name, kw = spec
    if name in ("Gaussian", "HetGaussian"):
        return rng.randn(n, 1) * 1.5
    if name == "Bernoulli":
        return (rng.rand(n, 1) < 0.5).astype(float)
    if name == "Poisson":
        return rng.poisson(3.0, size=(n, 1)).astype(float)
    if name in ("Gamma", "Exponential"):
        return rng.gamma(2.0, 1.0, size=(n, 1)) + 1e-3
    if name == "Beta":
        return np.clip(rng.beta(2.0, 3.0, size=(n, 1)), 1e-4, 1 - 1e-4)
    if name == "Categorical":
        return rng.randint(1, kw["K"] + 1, size=(n, 1)).astype(float)
    raise ValueError(name)


# --------------------------------------------------------------------------- G1
def gen_likelihoods(liks):
    cases = [("Gaussian", {"sigma": 0.5}), ("Gaussian", {"sigma": 1.0}), ("Bernoulli", {}), ("HetGaussian", {}),
             ("Poisson", {}), ("Exponential", {}), ("Gamma", {}), ("Beta", {}),
             ("Categorical", {"K": 3}), ("Categorical", {"K": 4}), ("Categorical", {"K": 5})]
    for k, spec in enumerate(cases):
        rng = np.random.RandomState(100 + k)
        n = 64
        df = dim_f(spec)
        y = sample_y(rng, spec, n)
        m = rng.uniform(-3, 3, size=(n, df))
        v = np.exp(rng.uniform(np.log(1e-3), np.log(4.0), size=(n, df)))
        # extreme rows: hit the clips (large |m|, large v)
        m[0, :] = 12.0
        m[1, :] = -12.0
        v[2, :] = 25.0
        m[3, :] = 25.0
        v[3, :] = 9.0
        m[4, :] = -30.0
        v[4, :] = 1e-6
        lik = make_lik(liks, spec)
        ve = lik.var_exp(y, m, v)
        dm, dv = lik.var_exp_derivatives(y, m, v)
        tag = spec[0].lower() + ("_K%d" % spec[1]["K"] if "K" in spec[1] else "") + \
            ("_s%g" % spec[1]["sigma"] if "sigma" in spec[1] else "")
        np.savez_compressed(os.path.join(OUT, "lik_%s.npz" % tag), spec=json.dumps(spec), y=y, m=m, v=v,
                            var_exp=np.asarray(ve).reshape(n, 1), var_exp_dm=np.asarray(dm).reshape(n, df),
                            var_exp_dv=np.asarray(dv).reshape(n, df))
        print("lik", tag, float(np.sum(ve)))


def gen_predictive(liks):
    """f2: `<likelihood>.predictive(m, v)` (predictive mean / variance of y).  `gh_T` records the Gauss-Hermite order
    the instance used: 20 on a fresh instance, 10 when `var_exp` ran first on it (GPy caches the first rule, quirk Q7)."""
    cases = [("Gaussian", {"sigma": 0.5}, False), ("Bernoulli", {}, False), ("HetGaussian", {}, False), ("Poisson", {}, False),
             ("Exponential", {}, False), ("Gamma", {}, False), ("Gamma", {}, True), ("Beta", {}, False), ("Beta", {}, True),
             ("Categorical", {"K": 3}, False), ("Categorical", {"K": 4}, False)]
    for k, (name, kw, cached) in enumerate(cases):
        rng = np.random.RandomState(500 + k)
        n = 48
        spec = (name, kw)
        df = dim_f(spec)
        m = rng.uniform(-2, 2, size=(n, df))
        v = np.exp(rng.uniform(np.log(1e-3), np.log(2.0), size=(n, df)))
        m[0, :], v[0, :] = 6.0, 4.0
        m[1, :], v[1, :] = -6.0, 1e-6
        lik = make_lik(liks, spec)
        gh_T = 10 if name == "Categorical" else 20
        if cached:
            lik.var_exp(sample_y(rng, spec, n), m, v)
            gh_T = 10
        if name == "Gaussian":
            mp, vp = lik.predictive(m, v, None)
        else:
            mp, vp = lik.predictive(m, v)
        tag = name.lower() + ("_K%d" % kw["K"] if "K" in kw else "") + ("_cached10" if cached else "")
        np.savez_compressed(os.path.join(OUT, "pred_%s.npz" % tag), spec=json.dumps(spec), m=m, v=v, gh_T=gh_T,
                            mean_pred=np.asarray(mp).reshape(n, -1), var_pred=np.asarray(vp).reshape(n, -1))
        print("pred", tag, float(np.sum(mp)), float(np.sum(vp)))


# --------------------------------------------------------------------------- G2
def inducing(M, P, Q, rng, jitter_blocks=True):
    if P == 1:
        base = np.linspace(0, 1, M)[:, None]
    else:
        g = int(np.ceil(M ** (1.0 / P)))
        grid = np.stack(np.meshgrid(*[np.linspace(0, 1, g)] * P, indexing="ij"), -1).reshape(-1, P)
        base = grid[:M]
    Z = np.tile(base, (1, Q))
    if jitter_blocks:
        h = spacing(M, P)
        Z = Z + 0.15 * h * rng.randn(*Z.shape)
    return Z


def spacing(M, P):
    return 1.0 / (M - 1) if P == 1 else M ** (-1.0 / P)


def gen_cov(stand, util):
    cases = [("well_M16", 16, 1, 3, (0.8, 1.0, 1.3)), ("well_M40_2d", 40, 2, 2, (0.8, 1.1)),
             ("ladder_M24", 24, 1, 2, (4.0, 6.0))]
    for k, (tag, M, P, Q, cs) in enumerate(cases):
        rng = np.random.RandomState(200 + k)
        Z = inducing(M, P, Q, rng, jitter_blocks=not tag.startswith("ladder"))
        h = spacing(M, P)
        var = 0.5 + 0.5 * rng.rand(Q)
        ell = np.array(cs) * h
        kern_list = util.latent_functions_prior(Q, lenghtscale=ell, variance=var, input_dim=P)
        rungs = []
        orig = stand.jitchol
        import GPy.util.linalg as gl
        gl.jitchol = lambda A, maxtries=5: orig(A, maxtries, _record=rungs)
        util.linalg.jitchol = gl.jitchol
        try:
            Kuu, Luu, Kuui = util.latent_funs_cov(Z, kern_list)
        finally:
            gl.jitchol = orig
            util.linalg.jitchol = orig
        np.savez_compressed(os.path.join(OUT, "cov_%s.npz" % tag), Z=Z, variance=var, lengthscale=ell, P=P,
                            Kuu=Kuu, Luu=Luu, Kuui=Kuui, rung=np.array(rungs))
        print("cov", tag, "rungs", rungs, "cond", [float(np.linalg.cond(Kuu[q])) for q in range(Q)])


# ----------------------------------------------------------------------- G3/G4/G5/G6
INF_CASES = [
    # tag, likelihood specs, N_t, M, Q, P, c (lengthscale/spacing), batch_scale?, kappa>0?
    ("notebook", [("Gaussian", {"sigma": 1.0}), ("Bernoulli", {})], [64, 47], 8, 2, 1, (1.0, 1.3), False, False),
    ("config1", [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})], [40, 33, 29], 16, 2, 1,
     (0.8, 1.2), False, False),
    ("config2", [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})], [64, 17, 40, 33],
     16, 3, 1, (0.8, 1.0, 1.3), False, False),
    ("config2_svi", [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})],
     [32, 32, 17, 32], 16, 3, 1, (0.8, 1.0, 1.3), True, True),
    ("config4", [("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {})], [17, 9, 17, 13],
     5, 4, 1, (0.8, 1.0, 1.3, 1.1), True, False),
    ("config5_2d", [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], [31, 64], 16, 2, 2, (0.9, 1.2), False,
     False),
]


def build_case(rng, specs, Ns, M, Q, P, cs, use_bs, use_kappa):
    T = len(specs)
    X = [np.sort(rng.rand(n, P), axis=0) if P == 1 else rng.rand(n, P) for n in Ns]
    Y = [sample_y(rng, s, n) for s, n in zip(specs, Ns)]
    Z = inducing(M, P, Q, rng)
    h = spacing(M, P)
    var = 0.5 + 0.5 * rng.rand(Q)
    ell = np.array(cs) * h
    Df = sum(dim_f(s) for s in specs)
    W = np.stack([np.where(rng.rand(Df) < 0.5, 1.0, -1.0) * (0.5 + 0.5 * rng.randn(Df)) for _ in range(Q)])  # Q x Df
    kappa = (0.1 + 0.2 * rng.rand(Q, Df)) if use_kappa else np.zeros((Q, Df))
    m_u = 1.5 * rng.randn(M, Q)
    Lfull = np.stack([np.eye(M) * (0.6 + 0.4 * rng.rand(M)) + 0.05 * np.tril(rng.randn(M, M), -1) for _ in range(Q)])
    r, c = np.tril_indices(M)
    L_flat = np.stack([Lfull[q][r, c] for q in range(Q)], axis=1)  # (M(M+1)/2, Q) row-major tril
    bs = [float(3.0 + t) for t in range(T)] if use_bs else None
    return dict(X=X, Y=Y, Z=Z, variance=var, lengthscale=ell, W=W, kappa=kappa, m_u=m_u, L_flat=L_flat, batch_scale=bs)


def gen_inference(stand, inf, util, hl, liks):
    for k, (tag, specs, Ns, M, Q, P, cs, use_bs, use_kappa) in enumerate(INF_CASES):
        rng = np.random.RandomState(300 + k)
        c = build_case(rng, specs, Ns, M, Q, P, cs, use_bs, use_kappa)
        T = len(specs)
        lik_list = [make_lik(liks, s) for s in specs]
        likelihood = hl.HetLikelihood(lik_list)
        Y_metadata = likelihood.generate_metadata()
        Df = likelihood.num_output_functions(Y_metadata)
        kern_list = util.latent_functions_prior(Q, lenghtscale=c["lengthscale"], variance=c["variance"], input_dim=P)
        W_list = [c["W"][q][:, None].copy() for q in range(Q)]
        kappa_list = [c["kappa"][q].copy() for q in range(Q)]
        _, B_list = util.LCM(input_dim=P, output_dim=Df, rank=1, kernels_list=kern_list, W_list=W_list,
                             kappa_list=kappa_list)
        engine = inf.SVMOGPInf()
        elbo, grads, post, post_F = engine.inference(c["m_u"], c["L_flat"], c["X"], c["Y"], c["Z"], kern_list,
                                                     likelihood, B_list, Y_metadata, batch_scale=c["batch_scale"])
        # G3: q(f_d) ; G4: KL
        Kuu, Luu, Kuui = util.latent_funs_cov(c["Z"], kern_list)
        p_U = inf.pu(Kuu=Kuu, Luu=Luu, Kuui=Kuui)
        q_U = inf.qu(mu_u=c["m_u"], chols_u=c["L_flat"])
        f_index = Y_metadata["function_index"].flatten()
        out = {}
        for d in range(Df):
            Xt = c["X"][f_index[d]]
            qf = engine.calculate_q_f(X=Xt, Z=c["Z"], q_U=q_U, p_U=p_U, kern_list=kern_list, B=B_list, M=M,
                                      N=Xt.shape[0], Q=Q, D=Df, d=d)
            out["m_fd_%d" % d] = qf.m_fd
            out["v_fd_%d" % d] = qf.v_fd
        KL = engine.calculate_KL(q_U=q_U, p_U=p_U, M=M, Q=Q)
        out["KL"] = np.asarray(KL).reshape(1, 1)
        out["elbo"] = np.asarray(elbo).reshape(1, 1)
        for q in range(Q):
            out["dL_dmu_u_%d" % q] = grads["dL_dmu_u"][q]
            out["dL_dL_u_%d" % q] = grads["dL_dL_u"][q]
            out["dL_dKmm_%d" % q] = grads["dL_dKmm"][q]
            for d in range(Df):
                out["dL_dKmn_%d_%d" % (q, d)] = grads["dL_dKmn"][q][d]
                out["dL_dKdiag_%d_%d" % (q, d)] = grads["dL_dKdiag"][q][d]
        for t in range(T):
            out["X_%d" % t] = c["X"][t]
            out["Y_%d" % t] = c["Y"][t]
        np.savez_compressed(os.path.join(OUT, "inf_%s.npz" % tag), spec=json.dumps(specs), T=T, M=M, Q=Q, P=P, Df=Df,
                            Z=c["Z"], variance=c["variance"], lengthscale=c["lengthscale"], W=c["W"], kappa=c["kappa"],
                            m_u=c["m_u"], L_flat=c["L_flat"],
                            batch_scale=np.array(c["batch_scale"] if c["batch_scale"] else [1.0] * T),
                            f_index=f_index, d_index=Y_metadata["d_index"].flatten(), **out)
        print("inf", tag, "ELBO", float(elbo))


MODEL_CASES = [
    # tag, base inference case index, batch_size (None = full batch), vem_step, perturb live W (quirk Q3)
    ("notebook_full", 0, None, True, False),
    ("config1_full", 1, None, True, False),
    ("config2_full", 2, None, True, False),
    ("config2_staleW", 2, None, True, True),
    ("config4_full", 4, None, True, False),
    ("config5_2d_full", 5, None, True, False),
    ("config2_svi_E", 2, 16, True, False),
    ("config2_svi_M", 2, 16, False, False),
]


def gen_model(stand, util, hl, svmogp, liks):
    for k, (tag, base, batch_size, vem_step, staleW) in enumerate(MODEL_CASES):
        (_, specs, Ns, M, Q, P, cs, use_bs, use_kappa) = INF_CASES[base]
        rng = np.random.RandomState(400 + k)
        c = build_case(rng, specs, Ns, M, Q, P, cs, False, False)
        T = len(specs)
        likelihood = hl.HetLikelihood([make_lik(liks, s) for s in specs])
        Y_metadata = likelihood.generate_metadata()
        Df = likelihood.num_output_functions(Y_metadata)
        kern_list = util.latent_functions_prior(Q, lenghtscale=c["lengthscale"], variance=c["variance"], input_dim=P)
        W_list = [c["W"][q][:, None].copy() for q in range(Q)]
        # SVMOGP tiles a common M x P inducing set over q (svmogp.py:52); take block 0 of the case's Z
        Z0 = c["Z"][:, :P].copy()
        np.random.seed(1234 + k)
        import random
        random.seed(99 + k)
        model = svmogp.SVMOGP(X=c["X"], Y=c["Y"], Z=Z0, kern_list=kern_list, likelihood=likelihood,
                              Y_metadata=Y_metadata, batch_size=batch_size, W_list=W_list)
        # overwrite the random variational init (svmogp.py:66-69) and de-tile Z with the case's values
        model.q_u_means[...] = c["m_u"]
        model.q_u_chols[...] = c["L_flat"]
        model.Z[...] = c["Z"]
        W_live = c["W"].copy()
        if staleW:
            W_live = W_live + 0.3 * rng.randn(*W_live.shape)
            for q in range(Q):
                model.B_list[q].W[...] = W_live[q][:, None]
        model.vem_step = vem_step
        model.parameters_changed()
        out = {}
        for t in range(T):
            out["Xall_%d" % t] = c["X"][t]
            out["Yall_%d" % t] = c["Y"][t]
            out["Xbatch_%d" % t] = model.Xmulti[t]
            out["Ybatch_%d" % t] = model.Ymulti[t]
        np.savez_compressed(
            os.path.join(OUT, "model_%s.npz" % tag), spec=json.dumps(specs), T=T, M=M, Q=Q, P=P, Df=Df,
            Z=c["Z"], variance=c["variance"], lengthscale=c["lengthscale"], W=W_live, W0=c["W"],
            kappa=np.zeros((Q, Df)), m_u=c["m_u"], L_flat=c["L_flat"], stochastic=int(batch_size is not None),
            batch_size=-1 if batch_size is None else batch_size, vem_step=int(vem_step),
            batch_scale=np.array(model.batch_scale),
            elbo=np.asarray(model.log_likelihood()).reshape(1, 1),
            g_m_u=np.asarray(model.q_u_means.gradient), g_L_u=np.asarray(model.q_u_chols.gradient),
            g_Z=np.asarray(model.Z.gradient),
            g_variance=np.array([float(np.ravel(kq.variance.gradient)[0]) for kq in kern_list]),
            g_lengthscale=np.array([float(np.ravel(kq.lengthscale.gradient)[0]) for kq in kern_list]),
            g_W=np.stack([np.ravel(B.W.gradient) for B in model.B_list]),
            g_kappa=np.stack([np.ravel(B.kappa.gradient) for B in model.B_list]), **out)
        print("model", tag, "ELBO", float(model.log_likelihood()))


MPRED_CASES = [
    # tag, base inference case (likelihood mix, M, Q, P, c), rows per task, new points per task
    ("config2", 2, [14, 11, 12, 9], 7),
    ("config1", 1, [13, 10, 12], 6),
    ("config5_2d", 5, [16, 20], 9),
]


def spaced_inputs(rng, n, P, shift=0.0):
    """n inputs in the unit cube whose mutual distances stay near the grid spacing: the N x N K_ff blocks the predict
    routes factorise then need no jitter, so the fixture does not depend on a borderline ladder decision."""
    if P == 1:
        return (((np.arange(n) + 0.5 + shift) / n + 0.25 / n * (rng.rand(n) - 0.5)) % 1.0)[:, None]
    g = int(np.ceil(n ** (1.0 / P)))
    cells = np.stack(np.meshgrid(*[np.arange(g)] * P, indexing="ij"), -1).reshape(-1, P)[:n]
    return ((cells + 0.5 + shift) / g + 0.25 / g * (rng.rand(n, P) - 0.5)) % 1.0


def gen_model_predict(stand, util, hl, svmogp, liks):
    for k, (tag, base, Ns, n_new) in enumerate(MPRED_CASES):
        (_, specs, _, M, Q, P, cs, _, _) = INF_CASES[base]
        rng = np.random.RandomState(600 + k)
        c = build_case(rng, specs, Ns, M, Q, P, cs, False, False)
        T = len(specs)
        c["X"] = [spaced_inputs(rng, n, P) for n in Ns]
        if P == 1:
            c["X"] = [np.sort(x, axis=0) for x in c["X"]]
        Xnew = [spaced_inputs(rng, n_new, P, shift=0.37) for _ in range(T)]
        likelihood = hl.HetLikelihood([make_lik(liks, s) for s in specs])
        Y_metadata = likelihood.generate_metadata()
        Df = likelihood.num_output_functions(Y_metadata)
        kern_list = util.latent_functions_prior(Q, lenghtscale=c["lengthscale"], variance=c["variance"], input_dim=P)
        W_list = [c["W"][q][:, None].copy() for q in range(Q)]
        np.random.seed(4321 + k)
        import random
        random.seed(77 + k)
        model = svmogp.SVMOGP(X=c["X"], Y=c["Y"], Z=c["Z"][:, :P].copy(), kern_list=kern_list, likelihood=likelihood,
                              Y_metadata=Y_metadata, batch_size=None, W_list=W_list)
        model.q_u_means[...] = c["m_u"]
        model.q_u_chols[...] = c["L_flat"]
        model.Z[...] = c["Z"]                      # blocks differ per latent: _raw_predict's use of block 0 is visible
        model.parameters_changed()
        f_index = Y_metadata["function_index"].flatten()
        rungs = []
        stand.Posterior.rungs = rungs
        out = {}
        try:
            for d in range(Df):
                xn = Xnew[f_index[d]]
                out["pn_m_%d" % d], out["pn_v_%d" % d] = model.predictive_new(xn, output_function_ind=d)
                out["rf_m_%d" % d], out["rf_v_%d" % d] = model._raw_predict_f(xn, output_function_ind=d)
            out["rs_m_0"], out["rs_v_0"] = model._raw_predict_stochastic(Xnew[f_index[0]], output_function_ind=0)
            for q in range(Q):
                out["ru_m_%d" % q], out["ru_v_%d" % q] = model._raw_predict(Xnew[0], latent_function_ind=q)
            pm, pv = model.predictive(Xnew)
        finally:
            stand.Posterior.rungs = None
        for t in range(T):
            out["pm_%d" % t], out["pv_%d" % t] = np.asarray(pm[t]), np.asarray(pv[t])
            out["X_%d" % t], out["Y_%d" % t], out["Xnew_%d" % t] = c["X"][t], c["Y"][t], Xnew[t]
        np.savez_compressed(
            os.path.join(OUT, "mpred_%s.npz" % tag), spec=json.dumps(specs), T=T, M=M, Q=Q, P=P, Df=Df, Z=c["Z"],
            variance=c["variance"], lengthscale=c["lengthscale"], W=c["W"], kappa=np.zeros((Q, Df)), m_u=c["m_u"],
            L_flat=c["L_flat"], f_index=f_index, d_index=Y_metadata["d_index"].flatten(), rungs=np.array(rungs),
            batch_scale=np.ones(T),
            elbo=np.asarray(model.log_likelihood()).reshape(1, 1), **out)
        print("mpred", tag, "ELBO", float(model.log_likelihood()), "K_chol rungs", sorted(set(rungs)),
              "max |predictive_new - raw_predict_f| d=0", float(np.max(np.abs(out["pn_m_0"] - out["rf_m_0"]))))


# ----------------------------------------------------------------------- N1: the reference itself at real sizes
C4_MIX = [("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {}), ("Gaussian", {"sigma": 0.5}),
          ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
H_MIX = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
REF_CASES = [
    # tag, likelihood specs, N_t, M, Q, P, c (lengthscale / inducing spacing), batch_size (None = full batch), vem_step
    ("c1_exact", [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})], [1000, 1000, 1000], 50, 2, 1,
     (0.8, 1.2), None, True),                                   # BASELINE config 1 at its exact size (README.md:31)
    ("h_mix_M128", H_MIX, [400, 383, 417, 350], 128, 3, 1, (0.8, 1.0, 1.3), None, True),   # specialised 128-tile kernels
    ("c4_mix_M160", C4_MIX, [230, 200, 217, 160, 256, 190, 240, 129], 160, 4, 1, (0.8, 1.0, 1.3, 1.1), None, True),
    ("c5_2d_M144", [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], [420, 500], 144, 2, 2, (0.9, 1.2), None, True),
    ("h_mix_M128_svi_M", H_MIX, [400, 383, 417, 350], 128, 3, 1, (0.8, 1.0, 1.3), 96, False),  # minibatch M-step, scales
]


def gen_reference_real_sizes(stand, inf, util, hl, svmogp, liks):
    """Row N1 of VERDICT r3: the reference's own `SVMOGP.parameters_changed` (svmogp.py:85-166) and
    `SVMOGPInf.inference` (svmogp_inf.py:23-109) run at BASELINE config 1's real size and at multi-tile M."""
    import random
    import time
    for k, (tag, specs, Ns, M, Q, P, cs, batch_size, vem_step) in enumerate(REF_CASES):
        rng = np.random.RandomState(700 + k)
        c = build_case(rng, specs, Ns, M, Q, P, cs, False, False)
        T = len(specs)
        likelihood = hl.HetLikelihood([make_lik(liks, s) for s in specs])
        Y_metadata = likelihood.generate_metadata()
        Df = likelihood.num_output_functions(Y_metadata)
        kern_list = util.latent_functions_prior(Q, lenghtscale=c["lengthscale"], variance=c["variance"], input_dim=P)
        W_list = [c["W"][q][:, None].copy() for q in range(Q)]
        np.random.seed(2468 + k)
        random.seed(135 + k)
        model = svmogp.SVMOGP(X=c["X"], Y=c["Y"], Z=c["Z"][:, :P].copy(), kern_list=kern_list, likelihood=likelihood,
                              Y_metadata=Y_metadata, batch_size=batch_size, W_list=W_list)
        model.q_u_means[...] = c["m_u"]
        model.q_u_chols[...] = c["L_flat"]
        model.Z[...] = c["Z"]
        model.vem_step = vem_step
        captured = {}
        real_inference = model.inference_method.inference

        def spy(*a, **kw):
            r = real_inference(*a, **kw)
            captured["elbo"], captured["grads"] = r[0], r[1]
            return r
        model.inference_method.inference = spy
        t0 = time.time()
        model.parameters_changed()
        dt = time.time() - t0
        model.inference_method.inference = real_inference
        grads = captured["grads"]
        Xb, Yb = model.Xmulti, model.Ymulti
        Kuu, Luu, Kuui = util.latent_funs_cov(c["Z"], kern_list)
        p_U = inf.pu(Kuu=Kuu, Luu=Luu, Kuui=Kuui)
        q_U = inf.qu(mu_u=c["m_u"], chols_u=c["L_flat"])
        f_index = Y_metadata["function_index"].flatten()
        out = {}
        for d in range(Df):
            Xt = Xb[f_index[d]]
            qf = model.inference_method.calculate_q_f(X=Xt, Z=c["Z"], q_U=q_U, p_U=p_U, kern_list=kern_list,
                                                      B=model.B_list, M=M, N=Xt.shape[0], Q=Q, D=Df, d=d)
            out["m_fd_%d" % d] = qf.m_fd
            out["v_fd_%d" % d] = qf.v_fd
        out["KL"] = np.asarray(model.inference_method.calculate_KL(q_U=q_U, p_U=p_U, M=M, Q=Q)).reshape(1, 1)
        for q in range(Q):
            if batch_size is not None:      # full batch: identical to g_m_u / g_L_u below (svmogp.py:111-112), not stored twice
                out["dL_dmu_u_%d" % q] = grads["dL_dmu_u"][q]
                out["dL_dL_u_%d" % q] = grads["dL_dL_u"][q]
            if M <= 128 and batch_size is None:   # larger M: pinned through g_Z / g_variance / g_lengthscale (size)
                out["dL_dKmm_%d" % q] = grads["dL_dKmm"][q]
            for d in range(Df):      # row sums of the un-stored dL_dKmn blocks (M,) and the dL_dKdiag vectors' sums: cheap pins
                out["dL_dKmn_rowsum_%d_%d" % (q, d)] = np.sum(grads["dL_dKmn"][q][d], axis=1)
                out["dL_dKdiag_sum_%d_%d" % (q, d)] = np.sum(grads["dL_dKdiag"][q][d])
        for t in range(T):
            out["Xall_%d" % t], out["Yall_%d" % t] = c["X"][t], c["Y"][t]
            out["Xbatch_%d" % t], out["Ybatch_%d" % t] = Xb[t], Yb[t]
        np.savez_compressed(
            os.path.join(OUT, "ref_%s.npz" % tag), spec=json.dumps(specs), T=T, M=M, Q=Q, P=P, Df=Df,
            Z=c["Z"], variance=c["variance"], lengthscale=c["lengthscale"], W=c["W"], W0=c["W"],
            kappa=np.zeros((Q, Df)), m_u=c["m_u"], L_flat=c["L_flat"], stochastic=int(batch_size is not None),
            batch_size=-1 if batch_size is None else batch_size, vem_step=int(vem_step),
            batch_scale=np.array(model.batch_scale), f_index=f_index, d_index=Y_metadata["d_index"].flatten(),
            elbo=np.asarray(model.log_likelihood()).reshape(1, 1), elbo_inference=np.asarray(captured["elbo"]).reshape(1, 1),
            g_m_u=np.asarray(model.q_u_means.gradient), g_L_u=np.asarray(model.q_u_chols.gradient),
            g_Z=np.asarray(model.Z.gradient),
            g_variance=np.array([float(np.ravel(kq.variance.gradient)[0]) for kq in kern_list]),
            g_lengthscale=np.array([float(np.ravel(kq.lengthscale.gradient)[0]) for kq in kern_list]),
            g_W=np.stack([np.ravel(B.W.gradient) for B in model.B_list]),
            g_kappa=np.stack([np.ravel(B.kappa.gradient) for B in model.B_list]),
            reference_seconds=dt, **out)
        print("ref", tag, "ELBO", float(model.log_likelihood()), "reference parameters_changed: %.2f s" % dt,
              "cond(Kuu)", ["%.1e" % np.linalg.cond(Kuu[q]) for q in range(Q)])


# ----------------------------------------------------------------- J1: the reference itself where GPy's jitter ladder lives
C1_MIX = [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})]
LADDER_CASES = [
    # tag, likelihood specs, N_t, M, Q, lengthscale / inducing spacing (None: absolute lengthscale below), variances, input offset
    # (i) BASELINE config 1's shape with the notebook's OWN hyper-parameters (notebooks/demo.ipynb cell 7: ls = 0.05, var = 0.5,
    #     Z = linspace(0, 1, M)): l / h = 2.45, cond(K_uu) = 1.1e12 -- LAPACK's dpotrf still succeeds (rung -1)
    ("c1_notebook_ell", C1_MIX, [1000, 1000, 1000], 50, 2, None, (0.5, 0.5), 0.0),
    # (ii) headline likelihood mix, M = 128, l = 4 h: plain dpotrf fails, rung 0 (jitter 1e-6 mean diag) holds; cond 1e7
    ("h_mix_M128_ladder", H_MIX, [400, 383, 417, 350], 128, 3, 4.0, (0.5, 0.7, 0.9), 0.0),
    # (iii) un-centred inputs (x in [1e4, 1e4 + 1]): GPy's expanded distance |x|^2 + |z|^2 - 2 x.z loses 8 digits, K_uu is
    #     indefinite by 2e-6 -- rung 0 (5e-7) fails as well, rung 1 holds
    ("c1_offset_rung1", C1_MIX, [300, 280, 310], 50, 2, 4.0, (0.5, 0.5), 1.0e4),
]


def _textbook_cholesky(A):
    """Cholesky-Banachiewicz, one row at a time, every entry one dot product: a valid factorisation whose sums run in a
    different order than LAPACK's blocked dpotrf."""
    n = A.shape[0]
    L = np.zeros_like(A)
    for i in range(n):
        for j in range(i):
            L[i, j] = (A[i, j] - np.dot(L[i, :j], L[j, :j])) / L[j, j]
        d = A[i, i] - np.dot(L[i, :i], L[i, :i])
        if not d > 0.0:
            raise np.linalg.LinAlgError("textbook Cholesky: non-positive pivot %d" % i)
        L[i, i] = np.sqrt(d)
    return L


def gen_reference_ladder(stand, inf, util, hl, svmogp, liks):
    """Row J1 of VERDICT r4: the reference's own `SVMOGP.parameters_changed` (svmogp.py:85-166) where `jitchol` (util.py:197-199)
    decides the numbers.  Each case is run TWICE: as it is, and with every covariance entry the kernel returns (`RBF.K`, reached
    from util.py:161,197) moved by one unit in the last place (K (1 +- 2^-52), seeded) -- what a second implementation's exp()
    is entitled to differ by -- and a THIRD time with LAPACK's dpotrf inside `jitchol` replaced by a textbook row-by-row Cholesky
    (same matrix, same rung, same jitter: only the order of the factorisation's sums changes -- what a GPU factorisation
    differs in).  `sens_<key>` = the larger |difference| of the two: the reference's own sensitivity to rounding-level changes,
    the only yardstick another implementation can be held to where it exceeds 1e-5."""

    import random
    import time
    import GPy.util.linalg as gl
    for k, (tag, specs, Ns, M, Q, c_ell, var, offset) in enumerate(LADDER_CASES):
        rng = np.random.RandomState(900 + k)
        c = build_case(rng, specs, Ns, M, Q, 1, (1.0,) * Q, False, False)
        h = spacing(M, 1)
        c["Z"] = np.tile(np.linspace(0, 1, M)[:, None], (1, Q)) + offset
        c["X"] = [x + offset for x in c["X"]]
        c["lengthscale"] = np.array([0.05] * Q if c_ell is None else [c_ell * h] * Q)
        c["variance"] = np.array(var, float)
        T = len(specs)

        def run(Z, ulp_rng=None, alt_rungs=None):
            likelihood = hl.HetLikelihood([make_lik(liks, s) for s in specs])
            Y_metadata = likelihood.generate_metadata()
            Df = likelihood.num_output_functions(Y_metadata)
            kern_list = util.latent_functions_prior(Q, lenghtscale=c["lengthscale"], variance=c["variance"], input_dim=1)
            W_list = [c["W"][q][:, None].copy() for q in range(Q)]
            np.random.seed(97531 + k)
            random.seed(864 + k)
            model = svmogp.SVMOGP(X=c["X"], Y=c["Y"], Z=Z[:, :1].copy(), kern_list=kern_list, likelihood=likelihood,
                                  Y_metadata=Y_metadata, batch_size=None, W_list=W_list)
            model.q_u_means[...] = c["m_u"]
            model.q_u_chols[...] = c["L_flat"]
            model.Z[...] = Z
            rungs = []
            orig = stand.jitchol
            rec = lambda A, maxtries=5: orig(A, maxtries, _record=rungs)
            if alt_rungs is not None:       # the SAME rung as the plain run, factorised by the textbook algorithm
                def rec(A, maxtries=5):
                    r = alt_rungs[len(rungs) % Q]
                    rungs.append(r)
                    jit = 0.0 if r < 0 else np.diag(A).mean() * 1e-6 * 10.0 ** r
                    return _textbook_cholesky(A + np.eye(A.shape[0]) * jit)
            gl.jitchol = rec
            util.linalg.jitchol = rec
            plain_K = stand.RBF.K
            if ulp_rng is not None:
                stand.RBF.K = lambda self, X, X2=None: (lambda K: K * (1.0 + ulp_rng.choice([-1.0, 1.0], size=K.shape) * 2.0 ** -52))(
                    plain_K(self, X, X2))
            try:
                t0 = time.time()
                model.parameters_changed()
                dt = time.time() - t0
                Kuu, Luu, Kuui = util.latent_funs_cov(Z, kern_list)
            finally:
                gl.jitchol = orig
                util.linalg.jitchol = orig
                stand.RBF.K = plain_K
            rungs = rungs[:Q]                          # (latent_funs_cov factorises the Q prior covariances first, util.py:196-199)
            p_U = inf.pu(Kuu=Kuu, Luu=Luu, Kuui=Kuui)
            q_U = inf.qu(mu_u=c["m_u"], chols_u=c["L_flat"])
            f_index = Y_metadata["function_index"].flatten()
            out = {}
            for d in range(Df):
                Xt = model.Xmulti[f_index[d]]
                qf = model.inference_method.calculate_q_f(X=Xt, Z=Z, q_U=q_U, p_U=p_U, kern_list=kern_list, B=model.B_list,
                                                          M=M, N=Xt.shape[0], Q=Q, D=Df, d=d)
                out["m_fd_%d" % d] = qf.m_fd
                out["v_fd_%d" % d] = qf.v_fd
            out["KL"] = np.asarray(model.inference_method.calculate_KL(q_U=q_U, p_U=p_U, M=M, Q=Q)).reshape(1, 1)
            out["elbo"] = np.asarray(model.log_likelihood()).reshape(1, 1)
            out["g_m_u"] = np.asarray(model.q_u_means.gradient).copy()
            out["g_L_u"] = np.asarray(model.q_u_chols.gradient).copy()
            out["g_Z"] = np.asarray(model.Z.gradient).copy()
            out["g_variance"] = np.array([float(np.ravel(kq.variance.gradient)[0]) for kq in kern_list])
            out["g_lengthscale"] = np.array([float(np.ravel(kq.lengthscale.gradient)[0]) for kq in kern_list])
            out["g_W"] = np.stack([np.ravel(B.W.gradient) for B in model.B_list])
            out["g_kappa"] = np.stack([np.ravel(B.kappa.gradient) for B in model.B_list])
            cond = [float(np.linalg.cond(Luu[q] @ Luu[q].T)) for q in range(Q)]
            return out, rungs, dt, f_index, Y_metadata["d_index"].flatten(), Df, cond

        out, rungs, dt, f_index, d_index, Df, cond = run(c["Z"])
        out2, rungs2, _, _, _, _, _ = run(c["Z"], np.random.RandomState(950 + k))
        assert rungs2 == rungs, (tag, rungs, rungs2)        # a case whose rung flips with the last bit pins nothing
        out3, _, _, _, _, _, _ = run(c["Z"], None, rungs)
        sens = {"sens_" + key: np.maximum(np.abs(np.asarray(out[key]) - np.asarray(out2[key])),
                                          np.abs(np.asarray(out[key]) - np.asarray(out3[key]))).astype(np.float32) for key in out}
        print("   one-ulp K / textbook-Cholesky sensitivities:",
              {key: "%.1e / %.1e" % (float(np.max(np.abs(out[key] - out2[key]))) / (float(np.max(np.abs(out[key]))) + 1e-300),
                                     float(np.max(np.abs(out[key] - out3[key]))) / (float(np.max(np.abs(out[key]))) + 1e-300))
               for key in ("elbo", "g_m_u", "g_L_u", "g_Z", "g_variance", "g_lengthscale", "g_W", "g_kappa", "m_fd_0", "v_fd_0")})
        for t in range(T):
            out["Xall_%d" % t], out["Yall_%d" % t] = c["X"][t], c["Y"][t]
            out["Xbatch_%d" % t], out["Ybatch_%d" % t] = c["X"][t], c["Y"][t]
        np.savez_compressed(
            os.path.join(OUT, "lad_%s.npz" % tag), spec=json.dumps(specs), T=T, M=M, Q=Q, P=1, Df=Df,
            Z=c["Z"], variance=c["variance"], lengthscale=c["lengthscale"], W=c["W"], W0=c["W"],
            kappa=np.zeros((Q, Df)), m_u=c["m_u"], L_flat=c["L_flat"], stochastic=0, batch_size=-1, vem_step=1,
            batch_scale=np.ones(T), f_index=f_index, d_index=d_index, rungs=np.array(rungs, dtype=np.int64),
            cond_jittered=np.array(cond), reference_seconds=dt, **out, **sens)
        print("lad", tag, "rungs", rungs, "ELBO", float(out["elbo"]), "cond(Kuu + jitter) %s" % ["%.1e" % x for x in cond],
              "reference parameters_changed: %.2f s" % dt,
              "| 1-ulp sensitivity (max |d| / max |ref|):",
              {key: "%.1e" % (float(np.max(sens["sens_" + key])) / (float(np.max(np.abs(out[key]))) + 1e-300))
               for key in ("elbo", "g_m_u", "g_L_u", "g_Z", "g_variance", "g_lengthscale", "g_W", "g_kappa")})


def gen_svi_trajectory(stand, util, hl, svmogp, liks):
    """[r6] Row f1 of SURVEY 8: the reference's OWN SVI driver -- `util.vem_algorithm(model, stochastic=True)`
    (util.py:316-329) over its own `SVMOGP.stochastic_grad` / `new_batch` / `set_data` / `callback` (svmogp.py:168-217) and
    `draw_mini_slices` (util.py:52-72) -- run for 13 consecutive iterations (vem_iters = 12: the callback stops at n_iter > 12) on
    the config-2 mix with contiguous minibatches of 16 rows (the 17-row task yields a 16-row and a ONE-row batch in turn).  What the
    stand-in supplies underneath is the paramz surface (`_grads`, `optimizer_array`, `model['<regexp>'].fix()`, Logexp) and climin's
    Adadelta recurrence (oracle/gpy_standin.py, restated from SURVEY appendix A); everything that decides WHICH rows, WHICH gating
    and WHICH gradients an iteration sees is the reference's code.  Recorded per iteration: the slice bounds of every task, the
    (vem_step, ve_count) the evaluation ran under, model.log_likelihood() after it, the optimiser vector the gradient was taken at
    and that gradient; at the end `model.elbo` as the callback wrote it and the optimiser's final vector."""
    import random
    (_, specs, Ns, M, Q, P, cs, _, _) = INF_CASES[2]
    batch_size, vem_iters, step_rate = 16, 12, 0.01
    rng = np.random.RandomState(977)
    c = build_case(rng, specs, Ns, M, Q, P, cs, False, False)
    T = len(specs)
    likelihood = hl.HetLikelihood([make_lik(liks, s) for s in specs])
    Y_metadata = likelihood.generate_metadata()
    Df = likelihood.num_output_functions(Y_metadata)
    kern_list = util.latent_functions_prior(Q, lenghtscale=c["lengthscale"], variance=c["variance"], input_dim=P)
    W_list = [c["W"][q][:, None].copy() for q in range(Q)]
    np.random.seed(4321)
    random.seed(17)
    model = svmogp.SVMOGP(X=c["X"], Y=c["Y"], Z=c["Z"][:, :P].copy(), kern_list=kern_list, likelihood=likelihood,
                          Y_metadata=Y_metadata, batch_size=batch_size, W_list=W_list)
    model.q_u_means[...] = c["m_u"]
    model.q_u_chols[...] = c["L_flat"]
    model.Z[...] = c["Z"]
    model.parameters_changed()
    slices = [[] for _ in range(T)]

    def recording_slicer(gen, t):               # (observes what the reference's own generator yields; changes nothing)
        for sl in gen:
            slices[t].append((sl.start, min(sl.stop, Ns[t])))
            yield sl
    model.slicer_list = [recording_slicer(g_, t) for t, g_ in enumerate(model.slicer_list)]
    trace = []
    ref_stochastic_grad = model.stochastic_grad

    def recording_grad(x):
        before = (int(bool(model.vem_step)), int(model.ve_count))
        g_ = ref_stochastic_grad(x)
        trace.append(dict(before=before, after=(int(bool(model.vem_step)), int(model.ve_count)), x=np.array(x, copy=True),
                          g=np.array(g_, copy=True), elbo=float(np.ravel(model.log_likelihood())[0]),
                          batch_scale=list(model.batch_scale)))
        return g_
    model.stochastic_grad = recording_grad
    captured = {}
    ref_adadelta = sys.modules["climin"].Adadelta

    def capturing_adadelta(*a, **kw):            # (keeps a handle on the optimiser util.py:327 creates: its final `wrt`)
        captured["opt"] = ref_adadelta(*a, **kw)
        captured["x0"] = np.array(captured["opt"].wrt, copy=True)
        return captured["opt"]
    sys.modules["climin"].Adadelta = capturing_adadelta
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)      # svmogp.py:203 assigns a (1,) array to a scalar slot
            util.vem_algorithm(model, stochastic=True, vem_iters=vem_iters, step_rate=step_rate)
    finally:
        sys.modules["climin"].Adadelta = ref_adadelta
    n_it = len(trace)
    assert n_it == vem_iters + 1 and all(len(s_) == n_it for s_ in slices), (n_it, [len(s_) for s_ in slices])
    out = {}
    for t in range(T):
        out["Xall_%d" % t], out["Yall_%d" % t] = c["X"][t], c["Y"][t]
    np.savez_compressed(
        os.path.join(OUT, "svi_traj_config2.npz"), spec=json.dumps(specs), T=T, M=M, Q=Q, P=P, Df=Df,
        Z=c["Z"], variance=c["variance"], lengthscale=c["lengthscale"], W=c["W"], W0=c["W"], kappa=np.zeros((Q, Df)),
        m_u=c["m_u"], L_flat=c["L_flat"], batch_size=batch_size, vem_iters=vem_iters, step_rate=step_rate, momentum=0.9,
        free=json.dumps([n for n, p_, _ in model._leaves() if not p_.is_fixed]),
        slice_begin=np.array([[slices[t][i][0] for t in range(T)] for i in range(n_it)]),
        slice_end=np.array([[slices[t][i][1] for t in range(T)] for i in range(n_it)]),
        gate_before=np.array([tr["before"] for tr in trace]), gate_after=np.array([tr["after"] for tr in trace]),
        batch_scale=np.array([tr["batch_scale"] for tr in trace]),
        elbo_trace=np.array([tr["elbo"] for tr in trace]), x_eval=np.stack([tr["x"] for tr in trace]),
        g_eval=np.stack([tr["g"] for tr in trace]), x0=captured["x0"], x_final=np.array(captured["opt"].wrt, copy=True),
        model_elbo=np.asarray(model.elbo, dtype=float), **out)
    print("svi trajectory: %d iterations, gates %s, ELBO %.6f -> %.6f" % (
        n_it, "".join("E" if tr["before"][0] else "M" for tr in trace), trace[0]["elbo"], trace[-1]["elbo"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    stand, inf, util, hl, svmogp, liks = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "ref":          # only the real-size family (the others are unchanged)
        gen_reference_real_sizes(stand, inf, util, hl, svmogp, liks)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ladder":       # only the jitter-ladder family
        gen_reference_ladder(stand, inf, util, hl, svmogp, liks)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "svi":          # only the SVI trajectory
        gen_svi_trajectory(stand, util, hl, svmogp, liks)
        return
    gen_likelihoods(liks)
    gen_predictive(liks)
    gen_cov(stand, util)
    gen_inference(stand, inf, util, hl, liks)
    gen_model(stand, util, hl, svmogp, liks)
    gen_model_predict(stand, util, hl, svmogp, liks)
    gen_reference_real_sizes(stand, inf, util, hl, svmogp, liks)
    gen_reference_ladder(stand, inf, util, hl, svmogp, liks)
    gen_svi_trajectory(stand, util, hl, svmogp, liks)


if __name__ == "__main__":
    main()
