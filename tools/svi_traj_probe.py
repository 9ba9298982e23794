"""Probe: the reference driver's SVI loop (hetmogp_amd.vem_algorithm(stochastic=True)) on the C3 shape, printing per-iteration state.
usage: python tools/svi_traj_probe.py [N_all] [iters] [step_rate] [optZ 0|1]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hetmogp_amd as H
from hetmogp_amd.kern import RBF
from hetmogp_amd.synthetic import make_case

SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
N_all = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
optZ = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
B, M, Q, P = 8192, 1024, 3, 1
prm, X, Y = make_case(SPECS, [N_all] * 4, M=M, Q=Q, P=P, seed=20260932)
lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson(), H.Gamma()])
np.random.seed(1)
h = 1.0 / (M - 1)
kern = [RBF(P, variance=float(prm["variance"][q]), lengthscale=h) for q in range(Q)]
model = H.HetMOGP(X, [y[:, None] for y in Y], prm["Z"][:, :P].copy(), kern, lik, lik.generate_metadata(), batch_size=B)
if not optZ:
    model.Z.fix()
inner = model.callback


def cb(i, max_iter, **kw):
    stop = inner(i, max_iter, verbose=False)
    n = i["n_iter"]
    if n <= 12 or n % 10 == 0:
        o = model.last
        print("it %3d %s elbo %.6g cond %s rungs %s strict_now %s var %s |g| %.3g" % (
            n, "E" if model.vem_step else "M", float(model._log_marginal_likelihood[0, 0]), ["%.2g" % c for c in o["cond_est"]],
            o["rungs"], model._strict_now, ["%.4g" % float(k.variance[0]) for k in model.kern_list],
            float(np.max(np.abs(i["gradient"]))) if len(i["gradient"]) else 0.0), flush=True)
    return stop
model.callback = cb
t0 = time.perf_counter()
with warnings.catch_warnings():
    warnings.simplefilter("always")
    try:
        H.vem_algorithm(model, stochastic=True, vem_iters=iters - 1, step_rate=rate, verbose=False)
    except Exception as e:
        print("FAILED:", type(e).__name__, e)
print("wall %.2f s, %d evaluations (%d strict), switches %d" % (time.perf_counter() - t0, model.evaluations, model.strict_evaluations,
                                                             model.strict_switches))
