"""GPU: prediction at the model level against what the reference's OWN SVMOGP methods returned (tests/golden/mpred_*.npz,
captured by oracle/make_golden.py: predictive_new, _raw_predict_f, _raw_predict_stochastic, _raw_predict, predictive --
svmogp.py:219-351).  `hmogp_predict_f` is pinned to `predictive_new`; the facade's opt-in route="reference" to the
`_raw_predict_f` route the reference's `predictive` / `negative_log_predictive` take; and the two are shown to differ."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN, "mpred_*.npz")))


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def build(g):
    import hetmogp_amd as H
    specs = json.loads(str(g["spec"]))
    T, Q, P = int(g["T"]), int(g["Q"]), int(g["P"])
    likelihood = H.HetLikelihood([getattr(H, n)(**kw) for n, kw in specs])
    kern_list = H.latent_functions_prior(Q, lenghtscale=g["lengthscale"], variance=g["variance"], input_dim=P)
    model = H.SVMOGP(X=[g["X_%d" % t] for t in range(T)], Y=[g["Y_%d" % t] for t in range(T)], Z=g["Z"][:, :P].copy(),
                     kern_list=kern_list, likelihood=likelihood, Y_metadata=likelihood.generate_metadata(),
                     W_list=[g["W"][q][:, None].copy() for q in range(Q)])
    model.q_u_means[...] = g["m_u"]
    model.q_u_chols[...] = g["L_flat"]
    model.Z[...] = g["Z"]                    # blocks differ per latent
    model.parameters_changed()
    assert rel(model.log_likelihood(), g["elbo"]) < 1e-8
    return model, [g["Xnew_%d" % t] for t in range(T)]


@pytest.mark.parametrize("strict", [False, True], ids=["default", "strict"])
@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_predict_f_is_the_references_predictive_new(path, strict):
    """hmogp_predict_f (q(f_d) at new inputs from q(u)) == SVMOGP.predictive_new of the reference, for every function d.
    [r5] The variance is a difference (explained - prior) that the default path forms through the explicit C_q: 1e-7 of its scale was
    the tolerance of rounds 2-4.  Measured this round (tools/strict_tolerance_probe.py): it was the FORM, not the hardware -- the
    strict mode (the reference's solve-based form, svmogp_inf.py:214-218) is held to 1e-9 here, the default to 1e-8."""
    from hetmogp_amd.engine import Engine
    g = np.load(path)
    specs = json.loads(str(g["spec"]))
    T, Q, M, P, Df = int(g["T"]), int(g["Q"]), int(g["M"]), int(g["P"]), int(g["Df"])
    e = Engine(specs, Q, M, P, strict_qf=strict)
    e.set_data([g["X_%d" % t] for t in range(T)], [g["Y_%d" % t] for t in range(T)])
    out = e.elbo_grad(Z=g["Z"], m_u=g["m_u"], L_flat=g["L_flat"], variance=g["variance"], lengthscale=g["lengthscale"],
                      W=g["W"], kappa=g["kappa"])
    ref_elbo = float(np.asarray(g["elbo"]).ravel()[0])
    assert abs(out["elbo"] - ref_elbo) < 1e-8 * abs(ref_elbo)
    for d in range(Df):
        m, v = e.predict_f(g["Xnew_%d" % int(g["f_index"][d])])
        assert rel(m[:, d:d + 1], g["pn_m_%d" % d]) < (1e-9 if strict else 1e-8), d
        assert rel(np.abs(v[:, d:d + 1]), g["pn_v_%d" % d]) < (1e-9 if strict else 1e-8), (d, rel(np.abs(v[:, d:d + 1]), g["pn_v_%d" % d]))


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_facade_reference_route_matches_raw_predict_f(path):
    g = np.load(path)
    model, Xnew = build(g)
    f_index = g["f_index"]
    gap = 0.0
    for d in range(int(g["Df"])):
        xn = Xnew[int(f_index[d])]
        m, v = model._raw_predict_f(xn, output_function_ind=d)
        assert model.last_predict_rung == -1                                   # as in the fixture: no jitter needed
        assert m.shape == g["rf_m_%d" % d].shape and v.shape == g["rf_v_%d" % d].shape
        assert rel(m, g["rf_m_%d" % d]) < 1e-8 and rel(v, g["rf_v_%d" % d]) < 1e-7, d
        m2, v2 = model.predictive_new(xn, output_function_ind=d)
        assert rel(m2, g["pn_m_%d" % d]) < 1e-8 and rel(v2, g["pn_v_%d" % d]) < 1e-7, d
        gap = max(gap, rel(m2, m))
    assert gap > 1e-3                        # the default route is a different estimator: documented in INTEGRATION.md
    m, v = model._raw_predict_stochastic(Xnew[int(f_index[0])], output_function_ind=0)
    assert rel(m, g["rs_m_0"]) < 1e-8 and rel(v, g["rs_v_0"]) < 1e-7


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_facade_predictive_and_raw_predict_reference_route(path):
    g = np.load(path)
    model, Xnew = build(g)
    T, Q = int(g["T"]), int(g["Q"])
    pm, pv = model.predictive(Xnew, route="reference")
    for t in range(T):
        assert pm[t].shape == g["pm_%d" % t].shape
        assert rel(pm[t], g["pm_%d" % t]) < 1e-7, t
        assert np.max(np.abs(pv[t] - g["pv_%d" % t])) <= 1e-7 * max(1.0, np.max(np.abs(g["pv_%d" % t]))), t
    dm, _ = model.predictive(Xnew)                                             # default: predictive_new semantics
    assert max(rel(dm[t], g["pm_%d" % t]) for t in range(T)) > 1e-3
    for q in range(Q):
        m, v = model._raw_predict(Xnew[0], latent_function_ind=q, route="reference")
        assert rel(m, g["ru_m_%d" % q]) < 1e-8 and rel(v, g["ru_v_%d" % q]) < 1e-7, q
    m0, _ = model._raw_predict(Xnew[0], latent_function_ind=0)
    assert rel(m0, g["ru_m_0"]) < 1e-8                                         # block 0 is block 0 on either route
    if Q > 1:                                                                  # block q != block 0 here
        m1, _ = model._raw_predict(Xnew[0], latent_function_ind=1)
        assert rel(m1, g["ru_m_1"]) > 1e-6
    # negative_log_predictive runs on both routes (Monte-Carlo: finite, and the routes differ)
    specs = json.loads(str(g["spec"]))
    if all(n not in ("Gamma", "Beta") for n, _ in specs):                      # the reference defines none for these
        Yt = [g["Y_%d" % t][:Xnew[t].shape[0]] for t in range(T)]
        a = model.negative_log_predictive(Xnew, Yt, num_samples=200, seed=3, route="reference")
        b = model.negative_log_predictive(Xnew, Yt, num_samples=200, seed=3)
        assert np.isfinite(a) and np.isfinite(b) and a != b


def test_reference_route_refuses_large_tasks():
    g = np.load(FILES[0])
    model, Xnew = build(g)
    model.REFERENCE_ROUTE_MAX_ROWS = 4
    with pytest.raises(ValueError):
        model._raw_predict_f(Xnew[0], output_function_ind=0)
