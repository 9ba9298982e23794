"""CPU: host-side logic of the facade that needs no GPU -- parameter containers (paramz-like slices, fix/unfix, Logexp),
the contiguous minibatch slicer (util.py:52-72), the climin-style Adadelta recurrence (util.py:327), likelihood metadata
(het_likelihood.py:24-44, checked against the f_index / d_index recorded from the reference in the golden fixtures)."""
import glob
import json
import os

import numpy as np

from conftest import GOLDEN


def test_param_slices_write_through_and_fix():
    from hetmogp_amd.param import Param, match, logexp_f, logexp_finv, logexp_gradfactor
    p = Param("m_u", np.zeros((4, 3)))
    p[:, 1:2].gradient = np.arange(4.0)[:, None]              # svmogp.py:106 idiom
    assert np.array_equal(p.gradient[:, 1], np.arange(4.0)) and p.gradient[:, 0].sum() == 0
    p[...] = 2.0
    assert np.all(p.values == 2.0)
    q = Param("variance", [0.5], positive=True)
    grp = match([("SVMOGP.m_u", p), ("SVMOGP.kern_q0.variance", q)], ".*.variance")
    assert len(grp) == 1
    grp.fix()
    assert q.is_fixed and not p.is_fixed
    grp.unfix()
    assert not q.is_fixed
    th = np.array([1e-9, 0.3, 5.0, 50.0])
    assert np.allclose(logexp_f(logexp_finv(th)), th, rtol=1e-12)
    x = logexp_finv(th)
    fd = (logexp_f(x + 1e-6) - logexp_f(x - 1e-6)) / 2e-6
    assert np.allclose(fd, logexp_gradfactor(th), rtol=1e-5, atol=1e-9)


def test_minibatch_slicer_is_contiguous_and_ordered():
    from hetmogp_amd import util
    sl = util.mini_slices(10, 4)
    assert [(s.start, s.stop) for s in sl] == [(0, 4), (4, 8), (8, 12)]            # last slice is short when applied
    it = util.draw_mini_slices(10, 4)
    seq = [next(it) for _ in range(7)]
    assert [(s.start, s.stop) for s in seq] == [(0, 4), (4, 8), (8, 12)] * 2 + [(0, 4)]   # always in order (util.py:70)
    import random
    random.seed(3)
    one = list(util.draw_mini_slices(10, 4, with_replacement=True))                 # ONE random slice, then stops
    assert len(one) == 1 and (one[0].start, one[0].stop) in [(0, 4), (4, 8), (8, 12)]
    assert util.mini_slices(8, 4) == [slice(0, 4), slice(4, 8)] and util.mini_slices(0, 4) == []


def test_adadelta_matches_its_recurrence():
    from hetmogp_amd.util import Adadelta
    A = np.diag([1.0, 10.0])
    x = np.array([1.0, -2.0])
    opt = Adadelta(x, lambda w: A @ w, step_rate=0.1, decay=0.9, momentum=0.9, offset=1e-4)
    w, gms, sms, step = np.array([1.0, -2.0]), np.zeros(2), np.zeros(2), np.zeros(2)
    it = iter(opt)
    for _ in range(25):
        next(it)
        s1 = 0.9 * step
        w = w - s1
        g = A @ w
        gms = 0.9 * gms + 0.1 * g ** 2
        s2 = np.sqrt(sms + 1e-4) / np.sqrt(gms + 1e-4) * g * 0.1
        w = w - s2
        step = s1 + s2
        sms = 0.9 * sms + 0.1 * step ** 2
        assert np.allclose(x, w, rtol=0, atol=1e-15)          # updated in place, like climin on model.optimizer_array
    assert opt.n_iter == 25


def test_metadata_matches_reference_fixtures():
    import hetmogp_amd as H
    for path in sorted(glob.glob(os.path.join(GOLDEN, "inf_*.npz"))):
        g = np.load(path)
        specs = json.loads(str(g["spec"]))
        lik = H.HetLikelihood([getattr(H, n)(**kw) for n, kw in specs])
        md = lik.generate_metadata()
        assert np.array_equal(md["function_index"], g["f_index"]) and np.array_equal(md["d_index"], g["d_index"])
        assert lik.num_output_functions(md) == int(g["Df"])
        assert lik.specs() == [(n, ({"sigma": kw["sigma"]} if n == "Gaussian" else kw)) for n, kw in specs]
    assert H.Categorical(5).get_metadata() == (1, 4, 4) and H.Gamma().get_metadata() == (1, 2, 1)
    assert H.Gaussian().sigma == 0.5                            # gaussian.py:21-24 default


def test_model_construction_helpers():
    import hetmogp_amd as H
    np.random.seed(0)
    W, kap = H.random_W_kappas(3, 5, rank=1)
    assert len(W) == 3 and W[0].shape == (5, 1) and all(np.all(k == 0) for k in kap)    # util.py:92-103
    ks = H.latent_functions_prior(2, lenghtscale=[0.1, 0.2], variance=[1.0, 2.0], input_dim=1)
    assert [k.name for k in ks] == ["kern_q0", "kern_q1"] and ks[1].lengthscale[0] == 0.2 and ks[1].variance.positive
    _, B = H.LCM(input_dim=1, output_dim=5, kernels_list=ks, W_list=W[:2], kappa_list=kap[:2], rank=1)
    assert len(B) == 2 and np.allclose(B[0].B, W[0] @ W[0].T)
    B[0].gradient = np.arange(10.0)
    assert np.array_equal(B[0].W.gradient.ravel(), np.arange(5.0)) and np.array_equal(B[0].kappa.gradient, np.arange(5.0, 10.0))
