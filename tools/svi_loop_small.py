#!/usr/bin/env python
"""40 iterations of the facade's device-resident SVI loop at the C3 size (N_all = 200 000 rows/task, batch 8192, M = 1024, Q = 3):
python tools/svi_loop_small.py [adadelta|natgrad]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import hetmogp_amd as H  # noqa: E402
from hetmogp_amd.kern import RBF  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "adadelta"
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
opt = model.device_adadelta(step_rate=0.005, momentum=0.9) if mode == "adadelta" else model.device_natgrad(gamma=0.1, step_rate=0.005, momentum=0.9)
it = iter(opt)
for _ in range(10):
    next(it)
t0 = time.perf_counter()
n = 30
for _ in range(n):
    next(it)
print("%s: %.3f ms / iteration" % (mode, 1e3 * (time.perf_counter() - t0) / n))
it.close()
