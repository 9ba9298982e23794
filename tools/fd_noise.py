"""Finite-difference noise of the eight_2d exact-mode case (tests/test_gpu_round2.py) on the small-model path."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from test_gpu_engine import make_engine, run, synth
specs = [("HetGaussian", {}), ("Categorical", {"K": 4}), ("Poisson", {}), ("Exponential", {})]
prm, prob, X, Y = synth(40 + len("eight_2d"), specs, [90, 80, 70, 60], 25, 3, 2, (1.0, 1.2, 0.9))
prm["kappa"] = 0.05 + 0.02 * np.arange(prob["Q"] * prob["Df"], dtype=float).reshape(prob["Q"], prob["Df"])
e = make_engine(prob, X, Y, quirks="exact")
out = run(e, prm)
out = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in out.items()}
rng = np.random.RandomState(5)
for key, gkey in (("m_u", "g_m_u"), ("L_flat", "g_L_u")):
    d = rng.randn(*np.shape(prm[key]))
    an = float(np.sum(np.asarray(out[gkey]) * d))
    for scale in (1e-6, 3e-6, 1e-5, 3e-5):
        eps = scale * np.abs(prm[key]).max()
        p1, p2 = dict(prm), dict(prm)
        p1[key], p2[key] = prm[key] + eps * d, prm[key] - eps * d
        fd = (run(e, p1)["elbo"] - run(e, p2)["elbo"]) / (2 * eps)
        print(key, "eps %.0e" % scale, "rel |fd - an| / |an| = %.2e" % (abs(fd - an) / abs(an)), "elbo", out["elbo"])
