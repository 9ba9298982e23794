#!/usr/bin/env python
"""Wall time of the SVI training loop (SVMOGP.stochastic_grad + Adadelta update) at the C3 size:
python tools/bench_facade.py [rows_per_task] [M] [batch] [device|host]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import hetmogp_amd as H  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402
from hetmogp_amd.util import Adadelta  # noqa: E402
from hetmogp_amd.kern import RBF  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
Q, P = 3, 1
specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
prm, X, Y = make_case(specs, [N] * 4, M=M, Q=Q, P=P, seed=3)
liks = [H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson(), H.Gamma()]
lik = H.HetLikelihood(liks)
meta = lik.generate_metadata()
np.random.seed(1)                   # the constructor draws W and the first batch (as the reference does): same model in both modes
kern = [RBF(P, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
model = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :P].copy(), kern_list=kern, likelihood=lik, Y_metadata=meta,
                 batch_size=B)
model[".*.lengthscale"].fix()      # as util.vem_algorithm does (util.py:284-331)
model[".*.kappa"].fix()
model.Z.fix()                       # optZ=False: 1024 inducing points one lengthscale apart do not survive raw SGD moves
model.stochastic = True
mode = sys.argv[4] if len(sys.argv) > 4 else "device"
if mode == "device":    # q(u) and its Adadelta accumulators resident in HBM (bit-identical iterates, tests/test_facade_gpu.py)
    opt = model.device_adadelta(step_rate=0.005, momentum=0.9)
else:                   # the reference-surface loop: climin-style Adadelta over model.optimizer_array on the host
    opt = Adadelta(model.optimizer_array, model.stochastic_grad, step_rate=0.005, momentum=0.9)
np.random.seed(0)                   # same minibatches in both modes: the final ELBO printed below must then agree exactly
it = iter(opt)
for _ in range(6):
    next(it)
t0 = time.perf_counter()
K = 20
for _ in range(K):
    next(it)
dt = (time.perf_counter() - t0) / K
print("SVI iteration [%s optimiser] (new batch + gradient + Adadelta update) N=%d M=%d batch=%d: %.2f ms = %.1f ELBO-steps/s of "
      "the training loop  ELBO %.6g" % (mode, N, M, B, 1e3 * dt, 1.0 / dt, float(model._log_marginal_likelihood[0, 0])))
ms, _ = model._engine.timings() if hasattr(model, "_engine") else ({}, {})
print({k: round(v, 2) for k, v in ms.items()})
