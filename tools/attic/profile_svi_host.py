#!/usr/bin/env python
"""Where does the host time of one SVI iteration go?  cProfile of the facade's DeviceAdadelta loop at the C3 size
(N_all = 200 000 rows/task resident, batch 8192, M = 1024, Q = 3):  python tools/profile_svi_host.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import hetmogp_amd as H  # noqa: E402
from hetmogp_amd.kern import RBF  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402

SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
N_all, B, M, Q, P = 200000, 8192, 1024, 3, 1
prm, X, Y = make_case(SPECS, [N_all] * 4, M=M, Q=Q, P=P, seed=3)
lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson(), H.Gamma()])
np.random.seed(1)
kern = [RBF(P, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
model = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :P].copy(), kern_list=kern, likelihood=lik,
                 Y_metadata=lik.generate_metadata(), batch_size=B)
model[".*.lengthscale"].fix()
model[".*.kappa"].fix()
model.Z.fix()
model.stochastic = True
opt = model.device_adadelta(step_rate=0.005, momentum=0.9)
it = iter(opt)
for _ in range(10):
    next(it)
dev = 0.0
t0 = time.perf_counter()
n = 50
for _ in range(n):
    next(it)
    dev += model._engine.timings()[0]["total"]
wall = 1e3 * (time.perf_counter() - t0) / n
print("wall %.3f ms / iteration, device (HIP events) %.3f ms -> host overhead %.3f ms" % (wall, dev / n, wall - dev / n))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    next(it)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
