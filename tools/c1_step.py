import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from hetmogp_amd.engine import Engine
from hetmogp_amd.synthetic import make_case
c1 = [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})]
prm, X, Y = make_case(c1, [1000] * 3, M=50, Q=2, P=1, seed=20260930)
e = Engine(c1, 2, 50, 1, reuse_outputs=True)
e.set_data(X, Y)
for _ in range(1500): e.elbo_grad(**prm)          # past the clock ramp (the first ~0.5 s after idle run slower)
t0 = time.perf_counter()
for _ in range(1000): out = e.elbo_grad(**prm)
dt = (time.perf_counter() - t0) / 1000
ms, nl = e.timings()
print("C1: %.1f us/step; device total %.1f us; graph (captures, replays) %s" % (1e6 * dt, 1e3 * ms["total"], e.graph_stats()))
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300): e.elbo_grad(**prm)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
