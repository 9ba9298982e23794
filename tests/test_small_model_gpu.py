"""GPU: the fused small-model kernels (hetmogp_amd/csrc/small_model.hip, M <= 64) and the small-problem mode against the REGULAR
kernels of the same library inside one process (HMOGP_CFG_NO_SMALL_PATH), on the reference-run fixtures and on seeded cases:
  * every output of one evaluation agrees to rounding (the two paths order their sums differently), for all three gradient gates;
  * BASELINE config 1 at its exact size: both paths against the reference's own numbers;
  * consumers that read what the evaluation left in HBM (posterior_u, predict_f, natgrad_step, the inner-protocol export) give the
    same answers behind either path;
  * a K_uu that needs GPy's jitter ladder: the small path detects it and the evaluation is repeated on the regular path (same rung
    as LAPACK), a forced rung is honoured by the small path itself, a failing forced rung raises LinAlgError;
  * M = 64 (the largest small size: four 64 x 66 matrices in LDS), P = 2, Q = 4, and M = 65 (first size of the regular path)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity

pytestmark = pytest.mark.gpu
KEYS = ["elbo", "KL", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"]


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def _pair(prob, X, Y):
    from hetmogp_amd.engine import Engine
    es = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"])
    er = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"], small_path=False)
    es.set_data(X, Y), er.set_data(X, Y)
    return es, er


def _args(prm, bs=None, **kw):
    a = dict(Z=prm["Z"], m_u=prm["m_u"], L_flat=prm["L_flat"], variance=prm["variance"], lengthscale=prm["lengthscale"],
             W=prm["W"], kappa=prm["kappa"], W0=prm.get("W0"), batch_scale=bs)
    a.update(kw)
    return a


@pytest.mark.parametrize("name", ["ref_c1_exact.npz", "inf_config1.npz", "inf_config2_svi.npz", "inf_config4.npz",
                                  "inf_config5_2d.npz", "inf_notebook.npz"])
def test_small_path_equals_regular_path(name):
    from oracle import svmogp_oracle as so
    from hetmogp_amd import _lib
    g = np.load(os.path.join(GOLDEN, name))
    prm, prob, X, Y, bs = so.load_case(g)
    es, er = _pair(prob, X, Y)
    for mask in (_lib.GROUP_ALL, _lib.GROUP_QU, _lib.GROUP_HYPER | _lib.GROUP_Z):
        a, b = es.elbo_grad(**_args(prm, bs, group_mask=mask)), er.elbo_grad(**_args(prm, bs, group_mask=mask))
        assert a["rungs"] == b["rungs"] == [-1] * prob["Q"]
        for k in KEYS:
            assert rel(a[k], b[k]) < 1e-9, (name, mask, k, rel(a[k], b[k]))
    if name == "ref_c1_exact.npz":          # ... and both against the reference itself
        for out in (a, b):
            pass
        full_s, full_r = es.elbo_grad(**_args(prm, bs)), er.elbo_grad(**_args(prm, bs))
        for k in KEYS:
            if k in g.files:
                assert_parity(full_s[k], g[k], ("small", k))
                assert_parity(full_r[k], g[k], ("regular", k))
    # consumers behind the evaluation
    es.elbo_grad(**_args(prm, bs)), er.elbo_grad(**_args(prm, bs))
    (wv_s, wi_s), (wv_r, wi_r) = es.posterior_u(), er.posterior_u()
    assert rel(wv_s, wv_r) < 1e-10 and rel(wi_s, wi_r) < 1e-9
    xn = X[0][: min(16, X[0].shape[0])]
    (ms_, vs_), (mr_, vr_) = es.predict_f(xn), er.predict_f(xn)
    assert rel(ms_, mr_) < 1e-10 and rel(vs_, vr_) < 1e-9
    raw_s, raw_r = es.debug_raw_grads([x.shape[0] for x in X]), er.debug_raw_grads([x.shape[0] for x in X])
    for q in range(prob["Q"]):
        assert rel(raw_s["dL_dKmm"][q], raw_r["dL_dKmm"][q]) < 1e-9
    (m1, L1), (m2, L2) = es.natgrad_step(0.05), er.natgrad_step(0.05)
    assert rel(m1, m2) < 1e-8 and rel(L1, L2) < 1e-8
    es.close(), er.close()


def _synth(seed, specs, Ns, M, Q, P, cs):
    from test_gpu_engine import synth
    return synth(seed, specs, Ns, M, Q, P, cs)


@pytest.mark.parametrize("M,Q,P", [(64, 4, 1), (64, 2, 2), (33, 3, 3), (1, 1, 1), (65, 2, 1)])
def test_small_sizes_vs_oracle_and_regular(M, Q, P):
    from oracle import svmogp_oracle as so
    specs = [("Gaussian", {"sigma": 0.5}), ("Poisson", {}), ("Beta", {}), ("Categorical", {"K": 3})]
    prm, prob, X, Y = _synth(700 + M + P, specs, [300, 257, 129, 200], M, Q, P, (0.9, 1.1, 1.3, 1.0)[:Q])
    want = so.elbo_grad_fused(prm, prob, X, Y)
    es, er = _pair(prob, X, Y)
    a, b = es.elbo_grad(**_args(prm)), er.elbo_grad(**_args(prm))
    for k in KEYS:
        if k == "KL":
            continue
        tol = 1e-8 if P == 1 else 2e-7
        assert rel(a[k], want[k]) < tol, ("small vs oracle", k, rel(a[k], want[k]))
        assert rel(a[k], b[k]) < 1e-9, ("small vs regular", k, rel(a[k], b[k]))
    es.close(), er.close()


def test_jitter_ladder_falls_back_to_the_regular_path():
    from oracle import svmogp_oracle as so
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {})]
    prm, prob, X, Y = _synth(13, specs, [300, 200], 24, 2, 1, (4.0, 5.0))
    prm["Z"] = np.tile(np.linspace(0, 1, 24)[:, None], (1, 2))
    want = so.elbo_grad_fused(prm, prob, X, Y)
    assert min(want["rungs"]) >= 0
    es, er = _pair(prob, X, Y)
    free_s, free_r = es.elbo_grad(**_args(prm)), er.elbo_grad(**_args(prm))
    assert free_s["rungs"] == free_r["rungs"] == want["rungs"]
    for k in KEYS:
        assert np.array_equal(np.asarray(free_s[k]), np.asarray(free_r[k])), k      # literally the regular path, second time round
    # the three-call form takes the same detour inside step_begin
    es.step_begin(**_args(prm))
    three = es.step_finish()
    assert three["rungs"] == want["rungs"] and three["elbo"] == free_s["elbo"]
    # a forced rung is the small path's own business ...
    f_s, f_r = es.elbo_grad(**_args(prm, forced_rung=want["rungs"])), er.elbo_grad(**_args(prm, forced_rung=want["rungs"]))
    # (cond(K_uu + jitter) ~ 1e7, |C| ~ 1e12: agreement is conditioning-limited; yardstick = the distance between the oracle's own
    #  two float64 restatements at the same rung, as in test_gpu_engine.py::test_forced_jitter_rung_matches_oracle)
    lit = so.elbo_grad_literal(prm, prob, X, Y, forced_rungs=want["rungs"])
    for k in KEYS:
        if k == "KL":
            continue
        yard = max(1e-5, 10.0 * rel(want[k], lit[k]))
        assert rel(f_s[k], want[k]) < yard and rel(f_r[k], want[k]) < yard, (k, rel(f_s[k], want[k]), rel(f_r[k], want[k]), yard)
    # ... and a forced rung that fails is an error on both paths
    for e in (es, er):
        with pytest.raises(np.linalg.LinAlgError):
            e.elbo_grad(**_args(prm, forced_rung=[-1, -1]))
    # the engine is usable afterwards
    again = es.elbo_grad(**_args(prm))
    assert again["elbo"] == free_s["elbo"]
    es.close(), er.close()


def test_small_mode_switches_with_the_evaluated_rows():
    """M = 128 is small-problem MODE territory (one stream) while the evaluation has <= 65536 rows and regular beyond: alternating
    minibatch / full evaluations on one engine must agree with an engine that never leaves the regular mode."""
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {})]
    from hetmogp_amd.synthetic import make_case
    prm, X, Y = make_case(specs, [60000, 50000], M=128, Q=2, P=1, seed=5)
    from hetmogp_amd.engine import Engine
    es, er = Engine(specs, 2, 128, 1), Engine(specs, 2, 128, 1, small_path=False)
    es.set_data(X, Y), er.set_data(X, Y)
    for rb, re_ in (([0, 0], [60000, 50000]), ([1000, 2000], [9000, 10000]), ([0, 0], [60000, 50000]), ([50000, 40000], [60000, 50000])):
        a = es.elbo_grad(row_begin=rb, row_end=re_, **prm)
        b = er.elbo_grad(row_begin=rb, row_end=re_, **prm)
        for k in KEYS:
            assert rel(a[k], b[k]) < 1e-10, (rb, k, rel(a[k], b[k]))
    es.close(), er.close()
