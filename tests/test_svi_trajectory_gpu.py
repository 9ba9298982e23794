"""GPU: the facade's SVI driver against 13 consecutive iterations of the REFERENCE'S OWN loop (row f1 of SURVEY 8).

tests/golden/svi_traj_config2.npz (oracle/make_golden.py:gen_svi_trajectory) is the reference's `util.vem_algorithm(model,
stochastic=True, vem_iters=12, step_rate=0.01)` (util.py:316-329) -- its own `SVMOGP.stochastic_grad` / `new_batch` / `set_data` /
`callback` (svmogp.py:168-217) and `draw_mini_slices` (util.py:52-72) -- on the config-2 mix with contiguous minibatches of 16 rows (a
17-row task alternates 16-row and ONE-row batches; two tasks end on ragged batches).  paramz and climin underneath are the stand-in's
restatements ("paramz/climin-unpinned"); which rows, which E/M gate and which gradients every iteration sees is the
reference's own code.  Asserted per iteration, to 1e-8: the slice bounds of every task, the (vem_step, ve_count) gate, the ELBO, the
optimiser vector the gradient was taken at, the gradient; at the end the optimiser's vector and `model.elbo` as the callback wrote
it -- for the host loop (`util.Adadelta`) and for the device-resident loop (`DeviceAdadelta`: q(u) and its accumulators in HBM)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_norm

pytestmark = pytest.mark.gpu
TOL = 1e-8


def _model(g):
    from test_facade_gpu import build_model
    model = build_model(g, int(g["batch_size"]))
    assert model._strict_auto and model.strict_switches == 0          # (the constructor default; K_uu is well conditioned here)
    return model


def _rows(model):
    return [r[0] for r in model._rows], [r[1] for r in model._rows]


def test_fixture_is_the_reference_loop():
    g = np.load(os.path.join(GOLDEN, "svi_traj_config2.npz"))
    n = int(g["vem_iters"]) + 1
    assert g["elbo_trace"].shape == (n,) and g["x_eval"].shape[0] == n and g["slice_begin"].shape == (n, int(g["T"]))
    assert "".join("E" if v else "M" for v in g["gate_before"][:, 0]) == "EEEEMEEEEMEEE"        # svmogp.py:190-198
    assert np.array_equal(g["model_elbo"].ravel(), g["elbo_trace"])                               # callback, svmogp.py:203
    assert json.loads(str(g["free"]))[:3] == ["inducing inputs", "m_u", "L_u"]                   # link order, svmogp.py:71-75
    assert int(np.min(g["slice_end"] - g["slice_begin"])) == 1                                    # the one-row minibatch is in


def test_host_adadelta_loop_reproduces_the_reference_trajectory():
    import hetmogp_amd as H
    g = np.load(os.path.join(GOLDEN, "svi_traj_config2.npz"))
    model = _model(g)
    trace = []
    inner = model.stochastic_grad

    def recording(x):
        before = (int(bool(model.vem_step)), int(model.ve_count))
        x_at = np.array(x, copy=True)
        gr = inner(x)
        trace.append(dict(before=before, rows=_rows(model), x=x_at, g=np.array(gr, copy=True),
                          elbo=float(model.log_likelihood()[0, 0]), bs=list(model.batch_scale)))
        return gr
    model.stochastic_grad = recording
    made = {}
    real = H.util.Adadelta

    class Capturing(real):
        def __init__(self, *a, **kw):
            real.__init__(self, *a, **kw)
            made["opt"] = self
    H.util.Adadelta = Capturing
    try:
        H.vem_algorithm(model, stochastic=True, vem_iters=int(g["vem_iters"]), step_rate=float(g["step_rate"]),
                        device_optimizer=False)
    finally:
        H.util.Adadelta = real
    assert len(trace) == g["elbo_trace"].shape[0]
    for i, tr in enumerate(trace):
        assert tr["rows"][0] == list(g["slice_begin"][i]) and tr["rows"][1] == list(g["slice_end"][i]), (i, tr["rows"])
        assert tr["before"] == tuple(g["gate_before"][i]), (i, tr["before"])
        assert np.allclose(tr["bs"], g["batch_scale"][i], rtol=1e-15)
        assert abs(tr["elbo"] - g["elbo_trace"][i]) <= TOL * abs(g["elbo_trace"][i]), (i, tr["elbo"], g["elbo_trace"][i])
        assert rel_norm(tr["x"], g["x_eval"][i]) < TOL, (i, "x", rel_norm(tr["x"], g["x_eval"][i]))
        assert rel_norm(tr["g"], g["g_eval"][i]) < TOL, (i, "g", rel_norm(tr["g"], g["g_eval"][i]))
    assert rel_norm(made["opt"].wrt, g["x_final"]) < TOL
    assert rel_norm(model.elbo.ravel(), g["model_elbo"].ravel()) < TOL
    assert model.strict_switches == 0 and model.strict_evaluations == 0


def test_device_adadelta_loop_reproduces_the_reference_trajectory():
    """The same trajectory with q(u) -- 456 of the 522 optimiser entries -- and its Adadelta accumulators resident in HBM
    (hmogp_qu_load / hmogp_qu_adadelta): ELBO, gating and row ranges per iteration, the host-side part of the gradient, and the point
    the model is left at when the loop ends (the LAST EVALUATION, like the reference's model object: climin applies the closing
    half-step to its own vector only)."""
    g = np.load(os.path.join(GOLDEN, "svi_traj_config2.npz"))
    model = _model(g)
    model[".*.lengthscale"].fix()
    model[".*.kappa"].fix()
    model.elbo = np.empty((int(g["vem_iters"]) + 1, 1))
    opt = model.device_adadelta(step_rate=float(g["step_rate"]), momentum=float(g["momentum"]))
    assert opt is not None
    free = json.loads(str(g["free"]))
    M, Q, P = int(g["M"]), int(g["Q"]), int(g["P"])
    n_qu = M * Q + (M * (M + 1) // 2) * Q
    small = np.r_[np.arange(0, M * Q * P), np.arange(M * Q * P + n_qu, g["x_eval"].shape[1])]   # Z | variance, W (host side)
    assert free[1:3] == ["m_u", "L_u"]
    it = iter(opt)
    for i in range(g["elbo_trace"].shape[0]):
        before = (int(bool(model.vem_step)), int(model.ve_count))
        info = next(it)
        assert info["n_iter"] == i + 1
        assert before == tuple(g["gate_before"][i]), (i, before)
        rb, re = _rows(model)
        assert rb == list(g["slice_begin"][i]) and re == list(g["slice_end"][i]), (i, rb, re)
        e = float(model._log_marginal_likelihood[0, 0])
        assert abs(e - g["elbo_trace"][i]) <= TOL * abs(g["elbo_trace"][i]), (i, e, g["elbo_trace"][i])
        gs = g["g_eval"][i][small]
        assert np.max(np.abs(info["gradient"] - gs)) <= TOL * (np.max(np.abs(g["g_eval"][i])) + 1e-300), (i, "small gradient")
        model.callback(info, max_iter=int(g["vem_iters"]), verbose=False)
    it.close()                                  # finish(): q(u) back into the model's arrays, at the last evaluation point
    assert rel_norm(model.elbo.ravel(), g["model_elbo"].ravel()) < TOL
    x_last = g["x_eval"][-1]
    assert rel_norm(model.q_u_means.values.ravel(), x_last[M * Q * P:M * Q * P + M * Q]) < TOL
    assert rel_norm(model.q_u_chols.values.ravel(), x_last[M * Q * P + M * Q:M * Q * P + n_qu]) < TOL
    assert rel_norm(model.Z.values.ravel(), x_last[:M * Q * P]) < TOL
