#!/usr/bin/env python
"""Step time of the other BASELINE.json configurations (single GPU; per-rank share for the 8-GPU config C4)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from hetmogp_amd.engine import Engine  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402

CASES = {
    "C1 demo (N=1000, M=50, Q=2, [HetGaussian,Bernoulli,Categorical3])":
        ([("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})], 1000, 50, 2, 1),
    "C2 (N=200k, M=512, Q=3, [Gaussian,Bernoulli,Poisson,Gamma])":
        ([("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})], 200000, 512, 3, 1),
    "C3 SVI minibatch (N_batch=8192, M=1024, Q=3)":
        ([("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})], 8192, 1024, 3, 1),
    "C4 per-rank share (N=125k of 1M, M=1024, Q=4, 8 likelihoods, Df=14)":
        ([("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {}), ("Gaussian", {"sigma": 0.5}),
          ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})], 125000, 1024, 4, 1),
    "C5 spatial 2-D (N=50k, M=2048, Q=2, [Categorical4,Gaussian])":
        ([("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], 50000, 2048, 2, 2),
}

ONLY = sys.argv[1] if len(sys.argv) > 1 else ""        # e.g. `python tools/bench_configs.py C4`
for name, (specs, N, M, Q, P) in CASES.items():
    if not name.startswith(ONLY):
        continue
    prm, X, Y = make_case(specs, [N] * len(specs), M=M, Q=Q, P=P, seed=1)
    e = Engine(specs, Q, M, P)
    e.set_data(X, Y)
    for _ in range(2):
        out = e.elbo_grad(**prm)
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        out = e.elbo_grad(**prm)
    dt = (time.perf_counter() - t0) / K
    ms, _ = e.timings()
    print("%-75s %8.2f ms/step  ELBO %.6g rungs %s  %s" % (name, 1e3 * dt, out["elbo"], out["rungs"],
                                                         {k: round(v, 2) for k, v in ms.items() if k != "total"}))
    if name.startswith("C5"):
        g = np.stack(np.meshgrid(np.linspace(0, 1, 256), np.linspace(0, 1, 256), indexing="ij"), -1).reshape(-1, 2)
        m, v = e.predict_f(g)                      # first call sizes the row workspaces for the grid
        t0 = time.perf_counter()
        m, v = e.predict_f(g)
        print("   predict_f on a 256x256 grid: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
    if name.startswith("C3"):                      # the 4 of 5 SVI updates that only move q(u) (svmogp.py:188-199)
        e.close()
        e = Engine(specs, Q, M, P, cache_kuu=True, reuse_outputs=True)
        e.set_data(X, Y)
        rng = np.random.RandomState(0)
        for i in range(8):
            if i == 3:
                t0 = time.perf_counter()
            prm["m_u"] = prm["m_u"] + 1e-3 * rng.randn(*prm["m_u"].shape)
            out = e.elbo_grad(group_mask=1, **prm)
        ms, _ = e.timings()
        print("   E-step (q(u) gradients only, K_uu chain cached): %.2f ms/step  %s"
              % (1e3 * (time.perf_counter() - t0) / 5, {k: round(v, 2) for k, v in ms.items() if k != "total"}))
    e.close()
