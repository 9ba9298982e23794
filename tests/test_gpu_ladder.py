"""GPU: the engine against runs of the REFERENCE ITSELF where GPy's jitter ladder decides the numbers (row J1 of VERDICT r4;
fixtures tests/golden/lad_*.npz from oracle/make_golden.py:gen_reference_ladder -- the reference's own SVMOGP.parameters_changed,
svmogp.py:85-166, over util.py:181-200's jitchol):

  lad_c1_notebook_ell     BASELINE config 1's shape (N_t=1000, M=50, Q=2) with the notebook's own hyper-parameters (demo.ipynb cell 7:
                          lengthscale 0.05, variance 0.5, Z = linspace): l/h = 2.45, cond(K_uu) = 1.1e12, LAPACK still succeeds (rung -1)
  lad_h_mix_M128_ladder   headline mix, M=128, l = 4 h: plain dpotrf fails, rung 0 holds (cond 1e7)
  lad_c1_offset_rung1     un-centred inputs (x in [1e4, 1e4+1]): K_uu indefinite by 2e-6, rung 0 fails too, rung 1 holds

What is asserted:
  * the FREE ladder (forced_rung = None) lands on the rung LAPACK took in the reference, in both modes;
  * STRICT q(f) mode (HMOGP_CFG_STRICT_QF): ELBO, KL, m_fd / v_fd and all seven gradient arrays against the reference's numbers
    under the element-wise criterion |a-b| <= 1e-5 |b| + 1e-9 max|b| (nothing loosened) AND an array norm of 1e-7 (1e-8 elsewhere
    in this suite: two valid K_uu^-1 of a cond-1e7 matrix differ by 1e-9, and g_W / g_Z reach 1.4e-8 of their scale through
    K_uu^-1 m) wherever the reference's own sensitivity to rounding-level changes (`sens_*` of the fixture: every covariance entry
    moved by one ulp, dpotrf replaced by a textbook Cholesky) is below 1e-7 of the array; where it is not (cond 1e12: the
    reference's own numbers move by 1e-4 when exp() rounds differently) within 50 x that sensitivity;
  * the DEFAULT mode (explicit C_q) is measured and printed beside it -- it is 1e-4..1e-3 off in g_W / g_kappa / g_Z there,
    which is why the strict mode exists."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, elementwise_excess, rel_norm

pytestmark = pytest.mark.gpu

KEYS = ["elbo", "KL", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"]
LAD = sorted(glob.glob(os.path.join(GOLDEN, "lad_*.npz")))


def _run(g, strict, forced=None, small_path=True):
    from hetmogp_amd.engine import Engine
    from oracle import svmogp_oracle as so
    prm, prob, X, Y, bs = so.load_case(g)
    e = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"], strict_qf=strict, small_path=small_path)
    e.set_data(X, Y)
    out = e.elbo_grad(Z=prm["Z"], m_u=prm["m_u"], L_flat=prm["L_flat"], variance=prm["variance"], lengthscale=prm["lengthscale"],
                      W=prm["W"], kappa=prm["kappa"], W0=prm.get("W0"), batch_scale=bs, forced_rung=forced)
    qf = {}
    for t in range(prob["T"]):
        m, v = e.predict_f(X[t])
        for d in range(prob["Df"]):
            if prob["f_index"][d] == t:
                qf["m_fd_%d" % d], qf["v_fd_%d" % d] = m[:, d], v[:, d]
    e.close()
    return out, qf, prob


def _errors(out, qf, g):
    rows = {}
    for k in KEYS:
        rows[k] = (rel_norm(out[k], g[k]), elementwise_excess(out[k], g[k]), _sens_ratio(g, k))
    for k in sorted(qf):
        rows[k] = (rel_norm(qf[k], g[k][:, 0]), elementwise_excess(qf[k], g[k][:, 0]), _sens_ratio(g, k))
    return rows


def _sens_ratio(g, k):
    return float(np.max(g["sens_" + k])) / (float(np.max(np.abs(g[k]))) + 1e-300)


def test_fixture_family_is_complete():
    names = {os.path.basename(p) for p in LAD}
    assert {"lad_c1_notebook_ell.npz", "lad_h_mix_M128_ladder.npz", "lad_c1_offset_rung1.npz"} <= names
    rungs = {os.path.basename(p): list(np.load(p)["rungs"]) for p in LAD}
    assert rungs["lad_c1_notebook_ell.npz"] == [-1, -1]
    assert rungs["lad_h_mix_M128_ladder.npz"] == [0, 0, 0]
    assert min(rungs["lad_c1_offset_rung1.npz"]) >= 1       # a case where rung 0 fails as well


@pytest.mark.parametrize("path", LAD, ids=os.path.basename)
@pytest.mark.parametrize("strict", [False, True], ids=["default", "strict"])
def test_free_ladder_lands_on_the_reference_rung(path, strict):
    g = np.load(path)
    out, _, _ = _run(g, strict)
    assert out["rungs"] == [int(r) for r in g["rungs"]]


@pytest.mark.parametrize("path", LAD, ids=os.path.basename)
def test_strict_mode_vs_reference_run_in_the_ladder_regime(path, capsys):
    g = np.load(path)
    out, qf, prob = _run(g, True)
    assert out["rungs"] == [int(r) for r in g["rungs"]]
    rows = _errors(out, qf, g)
    dout, dqf, _ = _run(g, False)
    drows = _errors(dout, dqf, g)
    with capsys.disabled():
        print("\n[ladder] %s  rungs %s  cond(K_uu + jitter) %s" % (os.path.basename(path), out["rungs"],
                                                                    ["%.1e" % c for c in g["cond_jittered"]]))
        print("[ladder]   %-12s %-24s %-24s %s" % ("array", "strict: norm / elem-excess", "default: norm / elem-excess", "reference 1-ulp sens"))
        for k, (rn, ex, sr) in rows.items():
            print("[ladder]   %-12s %9.2e / %-12.3g %9.2e / %-12.3g %9.2e" % (k, rn, ex, drows[k][0], drows[k][1], sr))
    for k, (rn, ex, sr) in rows.items():
        a = out[k] if k in out else qf[k]
        b = g[k] if k in out else g[k][:, 0]
        if sr < 1e-7:      # the reference's own numbers are stable to 1e-7 here: the element-wise criterion, nothing loosened
            assert rn < 1e-7, (k, "norm", rn)
            assert ex <= 1.0, (k, "element-wise excess", ex)
        else:              # conditioning-limited (the reference moves by `sr` when exp() rounds differently): 50 x its own sensitivity
            a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
            bound = 1e-5 * np.abs(b) + 1e-9 * np.max(np.abs(b)) + 50.0 * float(np.max(g["sens_" + k]))
            assert np.all(np.abs(a - b) <= bound), (k, "beyond 50 x the reference's own one-ulp sensitivity", rn, sr)


@pytest.mark.parametrize("name", ["lad_h_mix_M128_ladder.npz", "lad_c1_offset_rung1.npz"])
def test_facade_strict_qf_vs_reference_run_in_the_ladder_regime(name):
    """The drop-in surface -- SVMOGP(X, Y, Z, kern_list, likelihood, Y_metadata, W_list, strict_qf=True) / parameters_changed() /
    log_likelihood() -- where the reference's own jitchol took rung 0 / rung 1: the numbers its model object held afterwards."""
    from test_facade_gpu import build_model
    g = np.load(os.path.join(GOLDEN, name))
    model = build_model(g, None, strict_qf=True)
    model.parameters_changed()
    got = dict(elbo=model.log_likelihood(), g_m_u=model.q_u_means.gradient, g_L_u=model.q_u_chols.gradient, g_Z=model.Z.gradient,
               g_variance=[k.variance.gradient[0] for k in model.kern_list],
               g_lengthscale=[k.lengthscale.gradient[0] for k in model.kern_list],
               g_W=np.stack([B.W.gradient.ravel() for B in model.B_list]),
               g_kappa=np.stack([B.kappa.gradient.ravel() for B in model.B_list]))
    assert np.shape(got["elbo"]) == (1, 1)
    for k, v in got.items():
        assert rel_norm(v, g[k]) < 1e-7, (k, rel_norm(v, g[k]))
        assert elementwise_excess(v, g[k]) <= 1.0, (k, elementwise_excess(v, g[k]))


def test_strict_qf_auto_switches_where_the_engine_flags_and_only_there():
    """SVMOGP(strict_qf="auto"): the default path until the engine reports an ill-conditioned K_uu, then the SAME evaluation again in
    the strict mode (HMOGP_EVAL_STRICT_QF, per evaluation) and the following ones too -- the reference's numbers at jitter rung 0
    without the user knowing about modes; on BASELINE config 1 it never switches and returns the default path's numbers."""
    import warnings
    from hetmogp_amd.engine import Engine
    from oracle import svmogp_oracle as so
    from test_facade_gpu import build_model
    g = np.load(os.path.join(GOLDEN, "lad_h_mix_M128_ladder.npz"))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # auto mode must not warn here: it acts instead
        model = build_model(g, None, strict_qf="auto")
        model.parameters_changed()
    assert model.strict_switches == 1 and model._strict_now
    for k, v in dict(elbo=model.log_likelihood(), g_m_u=model.q_u_means.gradient, g_L_u=model.q_u_chols.gradient, g_Z=model.Z.gradient,
                     g_W=np.stack([B.W.gradient.ravel() for B in model.B_list]),
                     g_kappa=np.stack([B.kappa.gradient.ravel() for B in model.B_list])).items():
        assert elementwise_excess(v, g[k]) <= 1.0, (k, elementwise_excess(v, g[k]))
    c1 = np.load(os.path.join(GOLDEN, "ref_c1_exact.npz"))
    m1 = build_model(c1, None, strict_qf="auto")
    m1.parameters_changed()
    assert m1.strict_switches == 0 and not m1._strict_now
    assert rel_norm(m1.log_likelihood(), c1["elbo"]) < 1e-8
    # the per-evaluation flag on a default engine == an engine created strict, bit for bit; and back again
    prm, prob, X, Y, bs = so.load_case(g)
    rungs = [int(r) for r in g["rungs"]]
    ea = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"])
    eb = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"], strict_qf=True)
    ea.set_data(X, Y), eb.set_data(X, Y)
    d0 = {k: np.array(v, copy=True) for k, v in ea.elbo_grad(batch_scale=bs, forced_rung=rungs, **prm).items() if k in KEYS}
    a = ea.elbo_grad(batch_scale=bs, forced_rung=rungs, strict_qf=True, **prm)
    b = eb.elbo_grad(batch_scale=bs, forced_rung=rungs, **prm)
    for k in KEYS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    d1 = ea.elbo_grad(batch_scale=bs, forced_rung=rungs, **prm)
    for k in KEYS:
        assert np.array_equal(np.asarray(d1[k]), d0[k]), ("default path after a strict evaluation", k)
    ea.close(), eb.close()


def test_condition_estimate_and_flag_say_when_to_switch_modes():
    """hmogp_outputs.cond_est = variance max_i (K_uu^-1)_ii (a lower bound of cond(K_uu), 30-150x below it) and HMOGP_FLAG_ILL_CONDITIONED:
    raised by the default mode where it leaves element-wise 1e-5 (cond >= ~1e4), by the strict mode only beyond cond ~1e7."""
    import warnings
    from test_facade_gpu import build_model
    seen = {}
    for name in ("ref_c1_exact.npz", "lad_h_mix_M128_ladder.npz", "lad_c1_notebook_ell.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        for strict in (False, True):
            out, _, _ = _run(g, strict, forced=[int(r) for r in g["rungs"]] if "rungs" in g.files else None)
            seen[(name, strict)] = (out["ill_conditioned"], [float(c) for c in out["cond_est"]])
    assert seen[("ref_c1_exact.npz", False)][0] is False and seen[("ref_c1_exact.npz", True)][0] is False
    assert max(seen[("ref_c1_exact.npz", False)][1]) < 5e2
    assert seen[("lad_h_mix_M128_ladder.npz", False)][0] is True           # cond 1e7: the default path is 1e-4 off there ...
    assert seen[("lad_h_mix_M128_ladder.npz", True)][0] is False            # (strict: what jitter rung 0 leaves behind stays unflagged)
    assert 1e4 < max(seen[("lad_h_mix_M128_ladder.npz", True)][1]) < 1e7    # the estimate itself: cond 1e7 / (30 ... 150)
    assert seen[("lad_c1_notebook_ell.npz", True)][0] is True               # cond 1e12: flagged in either mode
    # both modes compute the estimate from their own K_uu^-1: same number to rounding
    a, b = seen[("lad_h_mix_M128_ladder.npz", False)][1], seen[("lad_h_mix_M128_ladder.npz", True)][1]
    assert np.allclose(a, b, rtol=1e-6)
    # the facade warns once, and says what to do
    g = np.load(os.path.join(GOLDEN, "lad_h_mix_M128_ladder.npz"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = build_model(g, None, strict_qf=False)   # (the constructor and every write to a parameter evaluate; the
        model.parameters_changed()                      #  explicit-inverse path by request: [r6] the default is "auto")
        model._dirty = True
        model.parameters_changed()
    assert model.last["ill_conditioned"] and max(model.last["cond_est"]) > 1e5
    msgs = [str(x.message) for x in w if "ill-conditioned" in str(x.message)]
    assert len(msgs) == 1 and "strict_qf=True" in msgs[0]
    assert model.strict_evaluations == 0 and model.evaluations >= 2


@pytest.mark.parametrize("name", ["lad_h_mix_M128_ladder.npz", "lad_c1_offset_rung1.npz"])
def test_north_star_constructor_default_meets_1e5_where_the_ladder_is_taken(name):
    """[r6] VERDICT r5 item 1a.  The model built EXACTLY as the north-star spells it -- HetMOGP(X, Y, Z, kern_list, likelihood,
    Y_metadata) plus the reference's own W_list keyword (svmogp.py:17; without it W is drawn at random) and NO engine option -- then
    model.parameters_changed() / model.log_likelihood(): element-wise 1e-5 against what the reference's model object held where its
    jitchol (util.py:197-199) took rung 0 / rung 1, without a warning.  The constructor default is strict_qf="auto"."""
    import json
    import warnings
    import hetmogp_amd as H
    g = np.load(os.path.join(GOLDEN, name))
    T, Q, P = int(g["T"]), int(g["Q"]), int(g["P"])
    likelihood = H.HetLikelihood([getattr(H, n)(**kw) for n, kw in json.loads(str(g["spec"]))])
    Y_metadata = likelihood.generate_metadata()
    kern_list = H.latent_functions_prior(Q, lenghtscale=g["lengthscale"], variance=g["variance"], input_dim=P)
    X, Y = [g["Xall_%d" % t] for t in range(T)], [g["Yall_%d" % t] for t in range(T)]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model = H.HetMOGP(X, Y, g["Z"][:, :P].copy(), kern_list, likelihood, Y_metadata,
                          W_list=[g["W0"][q][:, None].copy() for q in range(Q)])
        model.q_u_means[...] = g["m_u"]
        model.q_u_chols[...] = g["L_flat"]
        model.Z[...] = g["Z"]
        for q in range(Q):
            model.B_list[q].W[...] = g["W"][q][:, None]
        model.parameters_changed()
    assert model.strict_switches >= 1 and model._strict_now and model.strict_evaluations >= 1
    assert model.last["rungs"] == [int(r) for r in g["rungs"]]
    got = dict(elbo=model.log_likelihood(), g_m_u=model.q_u_means.gradient, g_L_u=model.q_u_chols.gradient, g_Z=model.Z.gradient,
               g_variance=[k.variance.gradient[0] for k in model.kern_list],
               g_lengthscale=[k.lengthscale.gradient[0] for k in model.kern_list],
               g_W=np.stack([B.W.gradient.ravel() for B in model.B_list]),
               g_kappa=np.stack([B.kappa.gradient.ravel() for B in model.B_list]))
    assert np.shape(got["elbo"]) == (1, 1)
    for k, v in got.items():
        assert rel_norm(v, g[k]) < 1e-7, (k, rel_norm(v, g[k]))
        assert elementwise_excess(v, g[k]) <= 1.0, (k, elementwise_excess(v, g[k]))


def test_strict_and_windows_are_refused_at_construction_and_default_resolves():
    """SVMOGP(strict_qf='auto', exact_zero_windows=True) used to construct, train, and raise at the first
    ill-conditioned evaluation.  Now an explicit request for both raises ValueError in the constructor; the default (None) resolves
    to 'auto' without windows and to False with them; exact_zero_windows='auto' yields to an explicit strict_qf."""
    from test_facade_gpu import build_model
    g = np.load(os.path.join(GOLDEN, "lad_h_mix_M128_ladder.npz"))
    for s in (True, "auto"):
        with pytest.raises(ValueError, match="exclude each other"):
            build_model(g, None, strict_qf=s, exact_zero_windows=True)
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = build_model(g, None, exact_zero_windows=True)
        m.parameters_changed()
    assert not m._strict_auto and not m.strict_qf and m.exact_zero_windows and m.strict_evaluations == 0
    assert any("exact_zero_windows is on" in str(x.message) for x in w)
    m2 = build_model(g, None, strict_qf="auto", exact_zero_windows="auto")
    assert m2._strict_auto and not m2.exact_zero_windows
    m3 = build_model(g, None)
    assert m3._strict_auto and not m3.exact_zero_windows


def test_default_mode_is_off_by_more_than_1e5_where_the_ladder_is_taken():
    """Documents WHY the strict mode exists: at rung 0 (cond 1e7) the explicit-inverse path is inside 1e-5 for the ELBO and the q(u)
    gradients and outside it for g_W / g_kappa / g_Z; if this ever stops failing the strict mode is no longer needed there."""
    g = np.load(os.path.join(GOLDEN, "lad_h_mix_M128_ladder.npz"))
    out, qf, _ = _run(g, False)
    rows = _errors(out, qf, g)
    assert rows["elbo"][1] <= 1.0 and rows["g_m_u"][1] <= 1.0
    assert max(rows[k][1] for k in ("g_W", "g_kappa", "g_Z")) > 1.0
    assert max(rows[k][0] for k in KEYS) < 1e-2          # ... and nowhere grossly wrong


def test_strict_equals_default_where_K_uu_is_well_conditioned():
    """Away from the ladder the two modes are the same numbers (1e-9): BASELINE config 1 at its exact size."""
    g = np.load(os.path.join(GOLDEN, "ref_c1_exact.npz"))
    a, qa, _ = _run(g, True)
    b, qb, _ = _run(g, False)
    for k in KEYS:
        assert rel_norm(a[k], b[k]) < 1e-9, k
        assert rel_norm(a[k], g[k]) < 1e-8, k
    for k in qa:
        assert rel_norm(qa[k], qb[k]) < 1e-9, k


@pytest.mark.parametrize("name", ["lad_h_mix_M128_ladder.npz", "lad_c1_offset_rung1.npz", "ref_h_mix_M128.npz"])
def test_strict_e_step_one_solve_form_vs_reference_run(name):
    """[r6] The strict mode's ONE-SOLVE form (E-steps and predictions: X = K^ Luu^-T only, the backward half of the reference's dpotrs
    moved onto M x M factors) against the reference's own runs where GPy's jitter ladder is taken: ELBO, KL and the q(u)
    gradients of an evaluation with group_mask = QU -- what `stochastic_grad` asks for in 4 of 5 SVI iterations (svmogp.py:188-199)
    and every L-BFGS iteration of a VE step (util.py:294-306) -- under the element-wise 1e-5 criterion; q(f) through predict_f; and
    the proof that it IS another form: its bundle's H slot holds X^T diag(beta) X, not A^T diag(beta) A."""
    from hetmogp_amd import _lib
    from hetmogp_amd.engine import Engine
    from oracle import svmogp_oracle as so
    g = np.load(os.path.join(GOLDEN, name))
    prm, prob, X, Y, bs = so.load_case(g)
    args = dict(Z=prm["Z"], m_u=prm["m_u"], L_flat=prm["L_flat"], variance=prm["variance"], lengthscale=prm["lengthscale"],
                W=prm["W"], kappa=prm["kappa"], W0=prm.get("W0"), batch_scale=bs)
    e = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"], strict_qf=True)
    e.set_data(X, Y)
    out = e.elbo_grad(group_mask=_lib.GROUP_QU, **args)
    if "rungs" in g.files:
        assert out["rungs"] == [int(r) for r in g["rungs"]]
    for k in ("elbo", "KL", "g_m_u", "g_L_u"):
        assert rel_norm(out[k], g[k]) < 1e-7, (k, rel_norm(out[k], g[k]))
        assert elementwise_excess(out[k], g[k]) <= 1.0, (k, elementwise_excess(out[k], g[k]))
    assert not np.any(out["g_Z"]) and not np.any(out["g_W"])
    for t in range(prob["T"]):
        m, v = e.predict_f(X[t])
        for d in range(prob["Df"]):
            if prob["f_index"][d] == t:
                assert elementwise_excess(m[:, d], g["m_fd_%d" % d][:, 0]) <= 1.0, ("m_fd", d)
                assert elementwise_excess(v[:, d], g["v_fd_%d" % d][:, 0]) <= 1.0, ("v_fd", d)
    # it IS the one-solve form: the bundle's H slot of latent 0 is the oracle's X^T diag(beta) X, not its A^T diag(beta) A -- for an
    # E-step and for a full-gradient evaluation alike (the condition estimate of these fixtures is below 1e6: DESIGN 13c)
    M, Q = prob["M"], prob["Q"]
    lay = so.stats_layout(prob)
    rungs = [int(r) for r in g["rungs"]] if "rungs" in g.files else None
    H = {}
    for form in ("one_solve", "two_solves"):
        sp = dict(prob, strict_qf=form)
        u = so.u_algebra(prm, sp, rungs)
        st, _ = so.local_stats(prm, sp, u, X, Y, bs)
        H[form] = st[lay["NG"]:lay["NG"] + M * M].reshape(M, M)
    assert rel_norm(np.tril(H["one_solve"]), np.tril(H["two_solves"])) > 1e-3
    outs = {}
    for mask in (_lib.GROUP_QU, _lib.GROUP_ALL):
        e.step_begin(group_mask=mask, **args)
        h = e.stats_read()[lay["NG"]:lay["NG"] + M * M].reshape(M, M)
        outs[mask] = e.step_finish()
        assert max(outs[mask]["cond_est"]) <= 1e6
        assert rel_norm(np.tril(h), np.tril(H["one_solve"])) < 1e-6, (mask, rel_norm(np.tril(h), np.tril(H["one_solve"])))
    for k in ("elbo", "g_m_u", "g_L_u"):
        assert rel_norm(outs[_lib.GROUP_QU][k], outs[_lib.GROUP_ALL][k]) < 1e-8, k
    e.close()
