#!/usr/bin/env python
"""Run a few steps of one configuration (for rocprofv3 traces):
python tools/run_config.py <rows_per_task> <M> <Q> [steps] [group_mask] [cache_kuu] [strict_qf]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hetmogp_amd.engine import Engine  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402

N, M, Q = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
mask = int(sys.argv[5]) if len(sys.argv) > 5 else 7
cache = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
strict = bool(int(sys.argv[7])) if len(sys.argv) > 7 else False
specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
prm, X, Y = make_case(specs, [N] * 4, M=M, Q=Q, P=1, seed=1)
e = Engine(specs, Q, M, 1, reuse_outputs=True, cache_kuu=cache, strict_qf=strict)
e.set_data(X, Y)
for _ in range(3):
    e.elbo_grad(group_mask=mask, **prm)
t0 = time.perf_counter()
for _ in range(steps):
    out = e.elbo_grad(group_mask=mask, **prm)
dt = (time.perf_counter() - t0) / steps
ms, _ = e.timings()
print("N=%d M=%d Q=%d mask=%d: %.3f ms/step  %s" % (N, M, Q, mask, 1e3 * dt, {k: round(v, 3) for k, v in ms.items()}))
