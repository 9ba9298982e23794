"""GPU: the natural-gradient update of q(u) (north-star) on the DEVICE-RESIDENT q(u) -- hmogp_qu_natgrad --
at the headline M = 1024.  The reference has no such step (it hands the Euclidean gradients of svmogp_inf.py:168-178 to
Adadelta), so these are property tests:
  * conjugate one-step optimum: Gaussian likelihoods and ONE latent GP make q(u)'s optimum closed-form; one natural-gradient
    step of size 1 lands on it (gradients vanish to rounding) from any starting point;
  * the in-place device step equals the host-returning step (hmogp_natgrad_step) bit for bit;
  * monotone ELBO for gamma <= 0.1 on the headline likelihood mix (non-conjugate), q(u) never leaving the device;
  * a step that would leave the positive-definite cone fails with LinAlgError and leaves q(u) untouched; a retry with a
    smaller gamma needs no new evaluation;
  * the facade's SVI loop with qu_optimizer="natgrad" (vem_algorithm) runs and improves the ELBO."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def _case(specs, N, M, Q, seed):
    from hetmogp_amd.synthetic import make_case
    prm, X, Y = make_case(specs, [N] * len(specs), M=M, Q=Q, P=1, seed=seed)
    return prm, X, Y


def test_conjugate_one_step_optimum_M1024_device_resident():
    from hetmogp_amd.engine import Engine
    from hetmogp_amd import _lib
    specs = [("Gaussian", {"sigma": 0.5}), ("Gaussian", {"sigma": 1.0})]
    M, Q = 1024, 1
    prm, X, Y = _case(specs, 6000, M, Q, 101)
    e = Engine(specs, Q, M, 1)
    e.set_data(X, Y)
    out0 = e.elbo_grad(**prm)
    small = {k: v for k, v in prm.items() if k not in ("m_u", "L_flat")}
    # host-returning step from the same evaluation (the reference point) ...
    m1, L1 = e.natgrad_step(1.0)
    # ... and the in-place step on the resident copy
    e.qu_load(prm["m_u"], prm["L_flat"])
    res0 = e.elbo_grad(m_u=None, L_flat=None, **small)
    assert res0["elbo"] == out0["elbo"]
    e.qu_natgrad(1.0)
    m1d, L1d = e.qu_read()
    assert np.array_equal(m1d, m1) and np.array_equal(L1d, L1)
    out1 = e.elbo_grad(m_u=None, L_flat=None, group_mask=_lib.GROUP_QU, **small)
    assert out1["elbo"] > out0["elbo"]
    chk = e.elbo_grad(m_u=m1, L_flat=L1, **small)            # gradients at the new point through the host path
    scale = max(np.max(np.abs(out0["g_m_u"])), np.max(np.abs(out0["g_L_u"])))
    assert np.max(np.abs(chk["g_m_u"])) < 1e-6 * scale and np.max(np.abs(chk["g_L_u"])) < 1e-6 * scale
    assert rel(chk["elbo"], out1["elbo"]) < 1e-12
    # a second full step stays at the optimum
    e.qu_load(m1, L1)
    e.elbo_grad(m_u=None, L_flat=None, **small)
    e.qu_natgrad(1.0)
    m2, L2 = e.qu_read()
    assert rel(m2, m1) < 1e-6 and rel(L2, L1) < 1e-6
    e.close()


def test_monotone_elbo_headline_mix_M1024_and_rejected_steps():
    from hetmogp_amd.engine import Engine
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
    M, Q = 1024, 3
    prm, X, Y = _case(specs, 4096, M, Q, 102)
    small = {k: v for k, v in prm.items() if k not in ("m_u", "L_flat")}
    e = Engine(specs, Q, M, 1)
    e.set_data(X, Y)
    e.qu_load(prm["m_u"], prm["L_flat"])
    prev = e.elbo_grad(m_u=None, L_flat=None, **small)["elbo"]
    for gamma in (0.1, 0.1, 0.05, 0.1):
        e.qu_natgrad(gamma)
        cur = e.elbo_grad(m_u=None, L_flat=None, **small)["elbo"]
        assert cur > prev, (gamma, cur, prev)
        prev = cur
    # an absurd step leaves the cone: LinAlgError, resident q(u) untouched, immediate retry works
    before = e.qu_read()
    with pytest.raises(np.linalg.LinAlgError):
        e.qu_natgrad(1e6)
    after = e.qu_read()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    e.qu_natgrad(0.05)                                           # no new evaluation in between
    assert e.elbo_grad(m_u=None, L_flat=None, **small)["elbo"] > prev
    # the same absurd step through the asynchronous entry point (ABI v7): refused ON THE DEVICE, nothing committed, reported later
    before = e.qu_read()
    e.qu_natgrad_async(1e6)
    assert e.qu_natgrad_status() is False
    after = e.qu_read()
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    e.elbo_grad(m_u=None, L_flat=None, **small)
    with pytest.raises(Exception):
        e.qu_natgrad(0.05)
        e.qu_natgrad(0.05)                                       # two steps from one evaluation: E_STATE
    e.close()


def test_facade_svi_loop_with_natural_gradient_e_steps():
    import hetmogp_amd as H
    from hetmogp_amd.kern import RBF
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {})]
    M, Q, N, B = 128, 2, 4000, 500
    prm, X, Y = _case(specs, N, M, Q, 103)

    def build():
        lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson()])
        np.random.seed(3)
        kern = [RBF(1, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
        return H.SVMOGP(X=X, Y=[y.reshape(-1, 1) for y in Y], Z=prm["Z"][:, :1].copy(), kern_list=kern, likelihood=lik,
                        Y_metadata=lik.generate_metadata(), batch_size=B)
    m_ng = build()
    H.vem_algorithm(m_ng, stochastic=True, vem_iters=40, step_rate=0.01, qu_optimizer="natgrad", natgrad_gamma=0.1)
    m_ad = build()
    H.vem_algorithm(m_ad, stochastic=True, vem_iters=40, step_rate=0.01)
    e_ng, e_ad = m_ng.elbo[:40, 0], m_ad.elbo[:40, 0]
    assert np.all(np.isfinite(e_ng)) and np.all(np.isfinite(e_ad))
    # natural-gradient E-steps climb much faster than Adadelta on the Euclidean gradient of q(u) from the same start
    assert np.mean(e_ng[-8:]) > np.mean(e_ng[:8]) and np.mean(e_ng[-8:]) > np.mean(e_ad[-8:])
    # q(u) came back to the host arrays when the loop ended and the model evaluates consistently from them
    m_ng.parameters_changed()
    assert np.isfinite(m_ng.log_likelihood()[0, 0])


def test_async_step_equals_synchronous_step_and_refused_steps_commit_nothing():
    """hmogp_qu_natgrad_async (ABI v7): committed on the device only inside the positive-definite cone; bit for bit the synchronous
    step; a refused step leaves the resident q(u) as it was; the next evaluation may be enqueued before the status is read.
    HMOGP_EVAL_NO_G_L: same ELBO, dL/dm and dL/dS, no factor gradient; Adadelta refuses to follow it."""
    from hetmogp_amd.engine import Engine
    from hetmogp_amd import _lib
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {})]
    M, Q = 256, 2
    prm, X, Y = _case(specs, 3000, M, Q, 104)
    small = {k: v for k, v in prm.items() if k not in ("m_u", "L_flat")}
    es = []
    for _ in range(2):
        e = Engine(specs, Q, M, 1)
        e.set_data(X, Y)
        e.qu_load(prm["m_u"], prm["L_flat"])
        es.append(e)
    a, b = es
    ra = a.elbo_grad(m_u=None, L_flat=None, group_mask=_lib.GROUP_QU, want_dL_dS=True, **small)
    rb = b.elbo_grad(m_u=None, L_flat=None, group_mask=_lib.GROUP_QU, want_dL_dS=True, skip_g_L=True, **small)
    assert ra["elbo"] == rb["elbo"] and np.array_equal(ra["dL_dS"], rb["dL_dS"])
    with pytest.raises(_lib.HetMOGPError):
        b.qu_adadelta(1, 0.01, 0.9, 0.9, 1e-4)                  # no gradient of the factor to feed it
    a.qu_natgrad(0.1)
    b.qu_natgrad_async(0.1)
    nb = b.elbo_grad(m_u=None, L_flat=None, group_mask=_lib.GROUP_QU, **small)     # enqueued BEFORE the status is read
    assert b.qu_natgrad_status() is True
    na = a.elbo_grad(m_u=None, L_flat=None, group_mask=_lib.GROUP_QU, **small)
    assert na["elbo"] == nb["elbo"] > ra["elbo"]
    qa, qb = a.qu_read(), b.qu_read()
    assert np.array_equal(qa[0], qb[0]) and np.array_equal(qa[1], qb[1])
    # (refused steps: test_monotone_elbo_headline_mix_M1024_and_rejected_steps)
    # two pending steps are an error; a synchronous step resolves a pending one first
    b.qu_natgrad_async(0.05)
    with pytest.raises(_lib.HetMOGPError):
        b.qu_natgrad_async(0.05)
    assert b.qu_natgrad_status() is True
    for e in es:
        e.close()


def test_facade_natgrad_loop_overlapped_equals_synchronous():
    """DeviceNatGrad(overlap=True) (asynchronous steps, E-step evaluations without the factor gradient) walks through the same
    iterates as the synchronous loop while no step is refused."""
    import hetmogp_amd as H
    from hetmogp_amd.kern import RBF
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {})]
    M, Q, N, B = 128, 2, 4000, 500
    prm, X, Y = _case(specs, N, M, Q, 105)
    elbos = []
    for overlap in (False, True):
        lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson()])
        np.random.seed(3)
        kern = [RBF(1, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
        m = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :1].copy(), kern_list=kern, likelihood=lik,
                     Y_metadata=lik.generate_metadata(), batch_size=B)
        m[".*.lengthscale"].fix()
        m[".*.kappa"].fix()
        m.Z.fix()
        m.stochastic = True
        opt = m.device_natgrad(gamma=0.1, step_rate=0.01, overlap=overlap)
        it = iter(opt)
        tr, infos = [], []
        for _ in range(30):
            infos.append(next(it))
            tr.append(float(m._log_marginal_likelihood[0, 0]))
        it.close()
        assert opt.rejected == 0
        elbos.append(tr)
        # [r6] ADVICE r5: what an E-step's info dict says about its OWN update.  Synchronous: the outcome; overlapped: pending
        # (gamma / step_taken None, the enqueued step size under gamma_requested, the last READ outcome under last_resolved --
        # None on the very first E-step, where nothing has been resolved yet)
        e_infos = [i for i in infos if i["step_taken"] is not None or i.get("pending")]
        assert len(e_infos) >= 20
        if overlap:
            assert all(i["gamma"] is None and i["step_taken"] is None and i["pending"] and i["gamma_requested"] > 0 for i in e_infos)
            assert e_infos[0]["last_resolved"] is None
            assert e_infos[2]["last_resolved"] == dict(gamma=e_infos[1]["gamma_requested"], step_taken=True)
        else:
            assert all(i["step_taken"] is True and i["gamma"] > 0 and "pending" not in i for i in e_infos)
    assert elbos[0] == elbos[1]
