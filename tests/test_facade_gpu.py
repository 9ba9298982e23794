"""GPU: the reference's model surface (SVMOGP / HetMOGP) over the engine -- constructor, log_likelihood(),
parameters_changed(), gradient gating in SVI mode, minibatch slicing, the VEM drivers -- against the golden
model_*.npz fixtures (outputs of the reference's own SVMOGP.parameters_changed)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def build_model(g, batch_size=None):
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
                     batch_size=batch_size, W_list=W_list)
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
