"""GPU: the reference's model surface (SVMOGP / HetMOGP) over the engine -- constructor, log_likelihood(),
parameters_changed(), gradient gating in SVI mode, minibatch slicing, the VEM drivers -- against the golden
model_*.npz fixtures (outputs of the reference's own SVMOGP.parameters_changed)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, elementwise_excess

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def build_model(g, batch_size=None, **model_kw):
    import hetmogp_amd as H
    specs = json.loads(str(g["spec"]))
    T, Q, P = int(g["T"]), int(g["Q"]), int(g["P"])
    liks = [getattr(H, n)(**kw) for n, kw in specs]
    likelihood = H.HetLikelihood(liks)
    md = likelihood.generate_metadata()
    kern_list = H.latent_functions_prior(Q, lenghtscale=g["lengthscale"], variance=g["variance"], input_dim=P)
    X = [g["Xall_%d" % t] for t in range(T)]
    Y = [g["Yall_%d" % t] for t in range(T)]
    W_list = [g["W0"][q][:, None].copy() for q in range(Q)]
    model = H.SVMOGP(X=X, Y=Y, Z=g["Z"][:, :P].copy(), kern_list=kern_list, likelihood=likelihood, Y_metadata=md,
                     batch_size=batch_size, W_list=W_list, **model_kw)
    model.q_u_means[...] = g["m_u"]
    model.q_u_chols[...] = g["L_flat"]
    model.Z[...] = g["Z"]
    for q in range(Q):
        model.B_list[q].W[...] = g["W"][q][:, None]
    return model


@pytest.mark.parametrize("tag", ["notebook_full", "config1_full", "config2_full", "config2_staleW", "config4_full",
                                 "config5_2d_full", "config2_svi_E", "config2_svi_M"])
def test_parameters_changed_matches_reference(tag):
    g = np.load(os.path.join(GOLDEN, "model_%s.npz" % tag))
    bs = int(g["batch_size"])
    model = build_model(g, None if bs < 0 else bs)
    model.vem_step = bool(g["vem_step"])
    model.parameters_changed()
    assert model.log_likelihood().shape == (1, 1)
    assert rel(model.log_likelihood(), g["elbo"]) < 1e-8
    assert rel(model.q_u_means.gradient, g["g_m_u"]) < 1e-8
    assert rel(model.q_u_chols.gradient, g["g_L_u"]) < 1e-8
    assert rel(model.Z.gradient, g["g_Z"]) < 1e-8
    assert rel([k.variance.gradient[0] for k in model.kern_list], g["g_variance"]) < 1e-8
    assert rel([k.lengthscale.gradient[0] for k in model.kern_list], g["g_lengthscale"]) < 1e-8
    assert rel(np.stack([B.W.gradient.ravel() for B in model.B_list]), g["g_W"]) < 1e-8
    assert rel(np.stack([B.kappa.gradient.ravel() for B in model.B_list]), g["g_kappa"]) < 1e-8
    assert np.allclose(model.batch_scale, g["batch_scale"])
    for got, key in ((model.q_u_means.gradient, "g_m_u"), (model.q_u_chols.gradient, "g_L_u"), (model.Z.gradient, "g_Z"),
                     (np.stack([B.W.gradient.ravel() for B in model.B_list]), "g_W")):
        assert elementwise_excess(got, g[key]) <= 1.0, (key, "element-wise 1e-5")


def test_optimizer_view_and_fd_of_objective():
    """optimizer_array / _grads (paramz semantics): finite differences of -log_likelihood in the optimiser's
    (Logexp-transformed) coordinates for the groups whose reference gradients are exact (m_u, L_u, lengthscale)."""
    g = np.load(os.path.join(GOLDEN, "model_notebook_full.npz"))
    model = build_model(g)
    model[".*.variance"].fix()
    model[".*.W"].fix()
    model[".*.kappa"].fix()
    model.Z.fix()
    x0 = model.optimizer_array.copy()
    g0 = model._grads(x0)
    rng = np.random.RandomState(0)
    d = rng.randn(x0.size)
    eps = 1e-6
    model.optimizer_array = x0 + eps * d
    f1 = model.objective_function()
    model.optimizer_array = x0 - eps * d
    f2 = model.objective_function()
    fd = (f1 - f2) / (2 * eps)
    assert abs(fd - g0 @ d) < 1e-4 * max(1.0, abs(fd))


def test_vem_full_batch_increases_elbo():
    """util.vem_algorithm (L-BFGS-B alternation, util.py:294-319) on the notebook-style toy: the ELBO must go up."""
    import hetmogp_amd as H
    g = np.load(os.path.join(GOLDEN, "model_notebook_full.npz"))
    model = build_model(g)
    e0 = float(model.log_likelihood()[0, 0])
    H.vem_algorithm(model, stochastic=False, vem_iters=1)
    e1 = float(model.log_likelihood()[0, 0])
    assert np.isfinite(e1) and e1 > e0 + 1.0


def test_svi_adadelta_runs_and_gates():
    """Stochastic mode: contiguous slices in order, batch_scale = N_all/N_batch, 4 E-steps then 1 M-step."""
    import hetmogp_amd as H
    g = np.load(os.path.join(GOLDEN, "model_config2_full.npz"))
    model = build_model(g, batch_size=16)
    assert model.stochastic and [x.shape[0] for x in model.Xmulti] == [16, 16, 16, 16]
    pattern = []
    x = model.optimizer_array
    for _ in range(10):
        pattern.append(model.vem_step)
        model.stochastic_grad(x)
    assert pattern == [True, True, True, True, False, True, True, True, True, False]
    np.random.seed(0)
    H.vem_algorithm(model, stochastic=True, vem_iters=30, step_rate=0.01)
    assert model.elbo.shape == (31, 1) and np.all(np.isfinite(model.elbo[:30]))


# ------------------------------------------------------------------------------------------------ round 2: wrappers
def _model_prm(model):
    Q = model.num_latent_funcs
    return dict(Z=model.Z.values, m_u=model.q_u_means.values, L_flat=model.q_u_chols.values,
                variance=np.array([float(k.variance[0]) for k in model.kern_list]),
                lengthscale=np.array([float(k.lengthscale[0]) for k in model.kern_list]),
                W=np.stack([np.ravel(B.W.values) for B in model.B_list]),
                kappa=np.stack([np.ravel(B.kappa.values) for B in model.B_list]))


def test_raw_predict_matches_woodbury_algebra():
    """SVMOGP._raw_predict (svmogp.py:219-253): mean = Kx^T woodbury_vector, var = Kdiag - diag(Kx^T woodbury_inv Kx),
    with woodbury_vector = Kuu^-1 m and woodbury_inv = Kuu^-1 - Kuu^-1 S Kuu^-1 = -C from the oracle's u_algebra."""
    from oracle import svmogp_oracle as so
    g = np.load(os.path.join(GOLDEN, "model_config2_full.npz"))
    model = build_model(g)
    prm = _model_prm(model)
    prob = so.make_problem(json.loads(str(g["spec"])), int(g["Q"]), int(g["M"]), int(g["P"]))
    u = so.u_algebra(prm, prob)
    Xnew = np.linspace(-0.1, 1.1, 57)[:, None]
    for q in range(prob["Q"]):
        mu, var = model._raw_predict(Xnew, latent_function_ind=q)
        Kx = so.rbf_K(prm["Z"][:, q:q + 1], Xnew, prm["variance"][q], prm["lengthscale"][q])
        assert rel(mu[:, 0], Kx.T @ u["a"][q]) < 1e-8
        want_var = np.abs(prm["variance"][q] + np.sum((u["C"][q] @ Kx) * Kx, 0))
        assert rel(var[:, 0], want_var) < 1e-8
        _, full = model._raw_predict(Xnew, latent_function_ind=q, full_cov=True)
        assert rel(np.diag(full), want_var) < 1e-8


def test_predictive_and_negative_log_predictive_wrappers():
    """SVMOGP.predictive (svmogp.py:333-351) = likelihood.predictive on q(f)(Xpred) of predict_f;
    SVMOGP.negative_log_predictive (svmogp.py:353-370) = - sum_t log_predictive: both against the per-likelihood
    building blocks fed with the oracle's q(f)."""
    from oracle import svmogp_oracle as so, likelihoods_oracle as lo
    from hetmogp_amd import engine as E
    g = np.load(os.path.join(GOLDEN, "model_config2_full.npz"))
    model = build_model(g)
    specs = json.loads(str(g["spec"]))
    T = int(g["T"])
    rng = np.random.RandomState(0)
    Xp = [np.sort(rng.rand(23 + 3 * t, 1), 0) for t in range(T)]
    mean, var = model.predictive(Xp)
    f_index = model.Y_metadata["function_index"].flatten()
    for t in range(T):
        m, v = model.predict_f(Xp[t])
        cols = [d for d in range(len(f_index)) if f_index[d] == t]
        name, kw = specs[t]
        want_m, want_v = lo.predictive(name, m[:, cols], np.abs(v[:, cols]), **kw)
        assert mean[t].shape == want_m.shape and rel(mean[t], want_m) < 1e-8 and rel(var[t], want_v) < 1e-8
    # NLPD: Gamma has no log_predictive in the reference -> use the Gaussian / Bernoulli / Poisson tasks of a second model
    g2 = np.load(os.path.join(GOLDEN, "model_notebook_full.npz"))
    m2 = build_model(g2)
    specs2 = json.loads(str(g2["spec"]))
    Xt = [np.sort(rng.rand(40, 1), 0) for _ in specs2]
    Yt = [rng.randn(40, 1) if n == "Gaussian" else (rng.rand(40, 1) < 0.5).astype(float) for n, _ in specs2]
    S = 4000
    nlpd = m2.negative_log_predictive(Xt, Yt, num_samples=S, seed=3)
    total = 0.0
    f2 = m2.Y_metadata["function_index"].flatten()
    for t, (name, kw) in enumerate(specs2):
        m, v = m2.predict_f(Xt[t])
        cols = [d for d in range(len(f2)) if f2[d] == t]
        rows = E.log_predictive_rows(name, Yt[t], m[:, cols], np.abs(v[:, cols]), S, 3 + t, **kw)
        total += (1.0 / S) * rows.sum()                     # the reference's 1/num_samples factor (bernoulli.py:144)
    assert np.isfinite(nlpd) and abs(nlpd + total) < 1e-12 * max(1.0, abs(total))


def test_natural_gradient_step_wrapper_improves_elbo():
    g = np.load(os.path.join(GOLDEN, "model_notebook_full.npz"))
    model = build_model(g)
    e0 = float(model.log_likelihood()[0, 0])
    m0 = model.q_u_means.values.copy()
    model.natural_gradient_step(gamma=0.1)
    e1 = float(model.log_likelihood()[0, 0])
    assert e1 > e0 and not np.array_equal(m0, model.q_u_means.values)


def test_foreign_set_data_then_minibatch_uses_the_full_data_again():
    """set_data(custom arrays) uploads them; the next new_batch()/stochastic_grad must evaluate row ranges of
    the ORIGINAL data (re-uploaded), reproducing the reference's SVI fixture."""
    g = np.load(os.path.join(GOLDEN, "model_config2_svi_E.npz"))
    bs = int(g["batch_size"])
    model = build_model(g, bs)
    T = int(g["T"])
    held_out = [np.sort(np.random.RandomState(t).rand(7, 1), 0) for t in range(T)]
    model.set_data(held_out, [np.ones((7, 1)) for _ in range(T)])
    model.parameters_changed()                                       # evaluates on the foreign data
    assert [r[1] for r in model._rows] == [7] * T and not model._engine_has_full
    model.slicer_list = [__import__("hetmogp_amd").util.draw_mini_slices(x.shape[0], bs) for x in model.Xmulti_all]
    model.set_data(*model.new_batch())                               # first contiguous slice again, as in the fixture
    assert model._engine_has_full
    model.vem_step = bool(g["vem_step"])
    model.parameters_changed()
    assert rel(model.log_likelihood(), g["elbo"]) < 1e-8
    assert rel(model.q_u_chols.gradient, g["g_L_u"]) < 1e-8


def test_direct_parameter_write_marks_the_model_dirty():
    """paramz re-evaluates on every parameter write; here a write marks the model dirty and the next read of a derived
    quantity (log_likelihood, predict_f, posteriors) re-evaluates."""
    g = np.load(os.path.join(GOLDEN, "model_notebook_full.npz"))
    model = build_model(g)
    e0 = float(model.log_likelihood()[0, 0])
    model.kern_list[0].lengthscale[:] = float(model.kern_list[0].lengthscale[0]) * 1.3
    assert model._dirty
    e1 = float(model.log_likelihood()[0, 0])
    assert not model._dirty and e1 != e0
    model.q_u_means *= 0.5
    assert model._dirty
    m, _ = model.predict_f(np.linspace(0, 1, 5)[:, None])
    assert not model._dirty and np.all(np.isfinite(m))


def test_fixed_groups_are_skipped_in_batch_mode():
    """Batch VEM E-steps fix every hyper-parameter: the engine then runs the q(u)-only evaluation (triangular fold);
    gradients_of_fixed=True restores the reference's habit of computing them anyway."""
    g = np.load(os.path.join(GOLDEN, "model_notebook_full.npz"))
    model = build_model(g)
    model.parameters_changed()
    full_gL = model.q_u_chols.gradient.copy()
    for grp in (".*.lengthscale", ".*.variance", ".*.W", ".*.kappa"):
        model[grp].fix()
    model.Z.fix()
    model.parameters_changed()
    assert rel(model.q_u_chols.gradient, full_gL) < 1e-9            # same q(u) gradient from the cheaper path
    assert all(float(k.variance.gradient[0]) == 0.0 for k in model.kern_list)
    model.gradients_of_fixed = True
    model.parameters_changed()
    assert any(float(k.variance.gradient[0]) != 0.0 for k in model.kern_list)


@pytest.mark.parametrize("n_iter", [50, 48])
def test_device_adadelta_iterates_are_bit_identical_to_host_adadelta(n_iter):
    """f1: SVI iterations (4 E-steps / 1 M-step gating, contiguous minibatches) from the state of the reference's
    model_config2_svi fixture: util.Adadelta on the host (climin's recurrence on model.optimizer_array) and
    DeviceAdadelta (q(u) + accumulators resident in HBM) must produce bit-identical ELBO traces and leave bit-identical
    parameters in the model -- the point of the last evaluation, as in the reference (the second half-step of the last
    update only exists in the optimiser's own vector).  50 iterations end on an M-step (q(u) did not move in it), 48 on
    an E-step (q(u) moved, the other parameters did not)."""
    import hetmogp_amd as H
    from hetmogp_amd.util import Adadelta
    g = np.load(os.path.join(GOLDEN, "model_config2_svi_E.npz"))
    bs = int(g["batch_size"])
    traces, finals = [], []
    for device in (False, True):
        model = build_model(g, bs)
        model[".*.lengthscale"].fix()
        model[".*.kappa"].fix()
        elbo = []
        if device:
            opt = model.device_adadelta(step_rate=0.01, momentum=0.9)
            assert opt is not None
        else:
            opt = Adadelta(model.optimizer_array, model.stochastic_grad, step_rate=0.01, momentum=0.9)

        def stop(info):
            elbo.append(float(model._log_marginal_likelihood[0, 0]))
            return info["n_iter"] >= n_iter
        opt.minimize_until(stop)
        traces.append(elbo)
        finals.append([model.q_u_means.values.copy(), model.q_u_chols.values.copy(), model.Z.values.copy()] +
                      [k.variance.values.copy() for k in model.kern_list] + [B.W.values.copy() for B in model.B_list])
    assert traces[0] == traces[1]
    for a, b in zip(*finals):
        assert np.array_equal(a, b)
    assert not np.array_equal(finals[0][0], g["m_u"])                 # and they did move


def test_vem_stochastic_uses_device_optimizer_and_syncs_q_u():
    import hetmogp_amd as H
    g = np.load(os.path.join(GOLDEN, "model_config2_full.npz"))
    model = build_model(g, batch_size=16)
    m0 = model.q_u_means.values.copy()
    H.vem_algorithm(model, stochastic=True, vem_iters=12, step_rate=0.01)
    assert not model._qu_on_device and not np.array_equal(model.q_u_means.values, m0)
    e = float(model.log_likelihood()[0, 0])                           # dirty after the sync: re-evaluated on read
    assert np.isfinite(e) and np.all(np.isfinite(model.elbo[:12]))


def test_q_u_read_mid_loop_sees_the_last_evaluation_and_observers_are_weak():
    """while a DeviceAdadelta loop owns q(u), `model.q_u_means` / `q_u_chols` read THROUGH to the device (as of the
    last evaluation) instead of showing the initial q(u); writes that bypass the Param class need `model.touch()`; a kernel
    reused in a second model neither keeps the first alive nor notifies it."""
    import gc
    import weakref
    from hetmogp_amd.util import Adadelta
    g = np.load(os.path.join(GOLDEN, "model_config2_svi_E.npz"))
    bs = int(g["batch_size"])
    host, dev = build_model(g, bs), build_model(g, bs)
    for m in (host, dev):
        m[".*.lengthscale"].fix()
        m[".*.kappa"].fix()
    seen_host, seen_dev = [], []
    oh = Adadelta(host.optimizer_array, host.stochastic_grad, step_rate=0.01, momentum=0.9)
    oh.minimize_until(lambda i: seen_host.append(host.q_u_means.values.copy()) or i["n_iter"] >= 7)
    od = dev.device_adadelta(step_rate=0.01, momentum=0.9)
    od.minimize_until(lambda i: seen_dev.append(dev.q_u_means.values.copy()) or i["n_iter"] >= 7)
    assert len(seen_host) == len(seen_dev) == 7
    for a, b in zip(seen_host, seen_dev):
        assert np.array_equal(a, b)                                    # what a callback sees mid-loop is the same q(u)
    assert not np.array_equal(seen_dev[0], seen_dev[-1])
    # a write around the Param class does not notify; touch() does
    e0 = float(dev.log_likelihood()[0, 0])
    dev.q_u_means.values[...] *= 1.01
    assert float(dev.log_likelihood()[0, 0]) == e0
    dev.touch()
    assert float(dev.log_likelihood()[0, 0]) != e0
    dev.q_u_means **= 1                                                # the added in-place operators notify
    assert dev._dirty
    # observers are weak references
    kern = host.kern_list
    ref = weakref.ref(host)
    del host, oh
    gc.collect()
    assert ref() is None
    kern[0].variance[...] = 0.7                                        # must not raise on the dead observer


def test_exact_zero_windows_auto_selects_by_input_layout_and_keeps_results():
    """exact_zero_windows="auto" (VERDICT r3 item 8): on for sorted 1-D inputs, off for unsorted or 2-D ones; results equal the
    dense model's either way (the mode only skips products with exact zeros)."""
    g = np.load(os.path.join(GOLDEN, "ref_h_mix_M128.npz"))
    import hetmogp_amd as H
    import hetmogp_amd.svmogp as sv
    dense = build_model(g)
    dense.parameters_changed()
    orig = sv.SVMOGP.__init__

    def patched(self, *a, **kw):
        kw["exact_zero_windows"] = "auto"
        return orig(self, *a, **kw)
    sv.SVMOGP.__init__ = patched
    try:
        auto = build_model(g)
        assert auto.exact_zero_windows is True            # the fixture's inputs are sorted per task (build_case)
        auto.parameters_changed()
        assert rel(auto.log_likelihood(), dense.log_likelihood()) < 1e-11
        assert rel(auto.q_u_chols.gradient, dense.q_u_chols.gradient) < 1e-10
        assert rel(auto.Z.gradient, dense.Z.gradient) < 1e-10
        g2 = np.load(os.path.join(GOLDEN, "ref_c5_2d_M144.npz"))
        assert build_model(g2).exact_zero_windows is False          # 2-D inputs
        gs = {k: g[k] for k in g.files}
        gs["Xall_0"] = g["Xall_0"][::-1].copy()
        gs["Yall_0"] = g["Yall_0"][::-1].copy()

        class _G(dict):
            files = list(gs.keys())
        assert build_model(_G(gs)).exact_zero_windows is False      # unsorted rows
    finally:
        sv.SVMOGP.__init__ = orig
    with pytest.raises(ValueError):
        H.SVMOGP(X=[g["Xall_0"]], Y=[g["Yall_0"]], Z=g["Z"][:, :1].copy(), kern_list=dense.kern_list[:1],
                 likelihood=H.HetLikelihood([H.Gaussian(sigma=0.5)]),
                 Y_metadata=H.HetLikelihood([H.Gaussian(sigma=0.5)]).generate_metadata(), exact_zero_windows="maybe")
