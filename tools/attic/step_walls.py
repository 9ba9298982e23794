import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from hetmogp_amd.engine import Engine, pinned_empty
from hetmogp_amd.synthetic import make_case
SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
prm, X, Y = make_case(SPECS, [200000] * 4, M=1024, Q=3, P=1, seed=20260929)
eng = Engine(SPECS, 3, 1024, 1, reuse_outputs=True)
eng.set_data(X, Y)
for k in ("Z", "m_u", "L_flat"):
    a = pinned_empty(np.shape(prm[k])); a[...] = prm[k]; prm[k] = a
for i in range(10):
    t0 = time.perf_counter()
    out = eng.elbo_grad(**prm)
    t1 = time.perf_counter()
    ms, _ = eng.timings()
    print(i, "wall %.2f ms  device %.2f ms" % (1e3 * (t1 - t0), ms["total"]), flush=True)
