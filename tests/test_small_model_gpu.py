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
        tol = 1e-8 if P == 1 else 2e-8
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


def test_graph_replay_tracks_parameter_values_and_keys():
    """[r4] hipGraph of the small-model evaluation: first call normal, second call captured, later calls replayed.  A replay must see
    NEW parameter values (they travel through the page-locked image the graph's upload node reads, and quad_kernel reads the mixing
    weights from device memory), a different gate / row range / forced rung is a different key, new data drops the graphs."""
    from oracle import svmogp_oracle as so
    from hetmogp_amd import _lib
    g = np.load(os.path.join(GOLDEN, "ref_c1_exact.npz"))
    prm, prob, X, Y, bs = so.load_case(g)
    es, er = _pair(prob, X, Y)
    rng = np.random.RandomState(0)
    outs = []
    for it in range(6):
        p2 = dict(prm)
        if it >= 2:                                   # every parameter group moves between replays
            p2["m_u"] = prm["m_u"] + 0.1 * it * rng.randn(*prm["m_u"].shape)
            p2["L_flat"] = prm["L_flat"] * (1.0 + 0.01 * it)
            p2["W"] = prm["W"] * (1.0 + 0.05 * it)
            p2["variance"] = prm["variance"] * (1.0 + 0.02 * it)
            p2["lengthscale"] = prm["lengthscale"] * (1.0 + 0.03 * it)
            p2["Z"] = prm["Z"] + 1e-3 * it
            p2["kappa"] = prm["kappa"] + 0.01 * it
        bs2 = [b * (1.0 + 0.5 * max(it - 1, 0)) for b in bs]
        a, b = es.elbo_grad(**_args(p2, bs2)), er.elbo_grad(**_args(p2, bs2))
        for k in KEYS:
            assert rel(a[k], b[k]) < 1e-9, (it, k, rel(a[k], b[k]))
        outs.append(a)
    cap, rep = es.graph_stats()
    assert cap == 1 and rep == 4, (cap, rep)          # call 0 normal, call 1 capture (+ launch), calls 2..5 replays
    assert er.graph_stats() == (0, 0)
    for k in KEYS:                                    # calls 0 (normal) and 1 (captured graph) had identical inputs
        assert np.array_equal(np.asarray(outs[0][k]), np.asarray(outs[1][k])), k
    # other keys: E-step gate, a row range, resident q(u)
    for kw in (dict(group_mask=_lib.GROUP_QU), dict(row_begin=[10, 20, 30], row_end=[900, 800, 1000])):
        for _ in range(3):
            a, b = es.elbo_grad(**_args(prm, bs, **kw)), er.elbo_grad(**_args(prm, bs, **kw))
            for k in KEYS:
                assert rel(a[k], b[k]) < 1e-9, (kw, k)
    assert es.graph_stats()[0] == 3
    es.qu_load(prm["m_u"], prm["L_flat"])
    small = {k: v for k, v in _args(prm, bs).items() if k not in ("m_u", "L_flat")}
    ref = er.elbo_grad(**_args(prm, bs))
    for i in range(4):
        a = es.elbo_grad(m_u=None, L_flat=None, **small)
        assert rel(a["elbo"], ref["elbo"]) < 1e-12 and rel(a["g_Z"], ref["g_Z"]) < 1e-9
        if i == 2:
            es.qu_adadelta(0, 0.01, 0.9, 0.9, 1e-4)   # an optimiser touching the resident q(u) between replays
            es.qu_adadelta(1, 0.01, 0.9, 0.9, 1e-4)
            m2, L2 = es.qu_read()
            ref = er.elbo_grad(**dict(_args(prm, bs), m_u=m2, L_flat=L2))
    assert es.graph_stats()[0] == 4
    # new data: graphs dropped, results follow the data
    es.set_data([x[:500] for x in X], [y[:500] for y in Y]), er.set_data([x[:500] for x in X], [y[:500] for y in Y])
    for _ in range(3):
        a, b = es.elbo_grad(**_args(prm, bs)), er.elbo_grad(**_args(prm, bs))
        assert rel(a["elbo"], b["elbo"]) < 1e-11 and rel(a["g_L_u"], b["g_L_u"]) < 1e-9
    assert es.graph_stats()[0] == 5 and es.graph_stats()[1] >= 8
    es.close(), er.close()


def test_quadrature_launch_variants_agree():
    """The one-launch quadrature of the small path (quad_multi_kernel, <= 8 segments and <= 2048 blocks per pool) against its two
    fall-backs -- per-segment launches for more tasks than a launch carries, and for more blocks -- and against the regular path;
    and a pool plan with several pools (chunk_rows smaller than the batch): the statistic bundle, zeroed once by u_small_kernel,
    accumulates over the pools."""
    from hetmogp_amd.engine import Engine
    # (a) 9 tasks -> per-segment launches on the small path
    specs9 = [("Gaussian", {"sigma": 0.4 + 0.05 * i}) if i % 3 == 0 else (("Bernoulli", {}) if i % 3 == 1 else ("Poisson", {}))
              for i in range(9)]
    prm, prob, X, Y = _synth(911, specs9, [60 + 7 * i for i in range(9)], 24, 2, 1, (1.0, 1.2))
    es, er = _pair(prob, X, Y)
    a, b = es.elbo_grad(**_args(prm)), er.elbo_grad(**_args(prm))
    for k in KEYS:
        assert rel(a[k], b[k]) < 1e-9, ("nine tasks", k, rel(a[k], b[k]))
    es.close(), er.close()
    # (b) a wave-per-row likelihood with more than 2048 quadrature blocks (9000 rows x 64 lanes / 256) beside a thread-per-row one
    specs = [("Categorical", {"K": 3}), ("Gaussian", {"sigma": 0.5})]
    prm, prob, X, Y = _synth(912, specs, [9000, 500], 32, 2, 1, (1.0, 1.2))
    es, er = _pair(prob, X, Y)
    a, b = es.elbo_grad(**_args(prm)), er.elbo_grad(**_args(prm))
    for k in KEYS:
        assert rel(a[k], b[k]) < 1e-9, ("many blocks", k, rel(a[k], b[k]))
    # (c) the same problem in pools of 2048 rows (five pools), one-launch quadrature in each
    ec = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"], chunk_rows=2048)
    ec.set_data(X, Y)
    c = ec.elbo_grad(**_args(prm))
    for k in KEYS:
        assert rel(c[k], b[k]) < 1e-9, ("several pools", k, rel(c[k], b[k]))
    es.close(), er.close(), ec.close()


def test_replayed_graph_detects_a_failed_factorisation():
    """A graph is captured on well-conditioned parameters; a later REPLAY gets inducing inputs whose K_uu needs GPy's jitter
    ladder.  The info words of a replay arrive with the results (the last block of finish_small_kernel writes them into the
    page-locked block) -- also when a three-call evaluation, whose info words travel by their own early copy, ran in between --
    and the evaluation must be repeated on the regular path: same rung and same numbers as the regular engine."""
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {})]
    prm, prob, X, Y = _synth(13, specs, [300, 200], 24, 2, 1, (4.0, 5.0))
    bad = dict(prm)
    bad["Z"] = np.tile(np.linspace(0, 1, 24)[:, None], (1, 2))      # (the ladder case of the test above)
    good = dict(prm)
    good["lengthscale"] = np.asarray(prm["lengthscale"]) * 0.02     # short lengthscale: K_uu close to diagonal
    good["Z"] = bad["Z"]
    es, er = _pair(prob, X, Y)
    for _ in range(3):                                               # normal, capture, replay
        a = es.elbo_grad(**_args(good))
    assert a["rungs"] == [-1, -1] and es.graph_stats() == (1, 1)
    es.step_begin(**_args(good))                                     # a three-call evaluation in between
    es.step_finish()
    a, b = es.elbo_grad(**_args(bad)), er.elbo_grad(**_args(bad))    # replay of the captured graph with a K_uu that fails
    assert min(b["rungs"]) >= 0 and a["rungs"] == b["rungs"]
    for k in KEYS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    a = es.elbo_grad(**_args(good))                                  # and the graph is still good for the next call
    c = er.elbo_grad(**_args(good))
    for k in KEYS:
        assert rel(a[k], c[k]) < 1e-9, k
    es.close(), er.close()
