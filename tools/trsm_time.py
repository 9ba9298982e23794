"""Time hmogp_potrs_rows-like solves through the engine's strict forward (one-solve E-step): prints forward_gemm ms.
usage: python tools/trsm_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hetmogp_amd.engine import Engine
from hetmogp_amd.synthetic import make_case
specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
prm, X, Y = make_case(specs, [200000] * 4, M=1024, Q=3, P=1, seed=1)
e = Engine(specs, 3, 1024, 1, reuse_outputs=True, strict_qf=True)
e.set_data(X, Y)
for _ in range(2):
    e.elbo_grad(group_mask=1, **prm)
ms, _ = e.timings()
print("E-step strict: trsm_solves %.2f ms (one forward solve)  T product %.2f  total %.2f" % (ms["trsm_solves"], ms["forward_gemm"], ms["total"]))
