#!/usr/bin/env python
"""Run a few steps of one BASELINE.json configuration through the C ABI (for rocprofv3 traces / PMC passes):
    python tools/run_workload.py <H|HE|HS|HSL|HSE|HD|C2|C3|C3E|C4|C5> [steps] [warmup]
H   headline: T=4 [Gaussian,Bernoulli,Poisson,Gamma], N_t=200000, M=1024, Q=3
C2  the same at M=512
C3  one full-gradient evaluation of an 8192-row minibatch out of N_all=1000000 resident rows per task (M=1024, Q=3)
C3E the E-step of the SVI loop on the same minibatch (q(u) group only, K_uu chain cached)
C4  one rank's share of config 4: 8 tasks x 125000 rows, M=1024, Q=4, Df=14
C5  2-D, T=2 [Categorical(4),Gaussian], N_t=50000, M=2048, Q=2
HD  headline shape with a DENSE-valued K^ (lengthscale = 40 inducing spacings, jitter rung 4 forced): no exact zeros
HE  the E-step of the headline workload (q(u) group only, K_uu chain cached): the fold-pair forward
HS  the headline workload in the strict q(f) mode (HMOGP_CFG_STRICT_QF), full gradients (one-solve form: estimate <= 1e6)
HSL the same shape at lengthscale = 4 inducing spacings, jitter rung 0 forced (estimate 5.5e5): where the mode is needed
HSE the same with group_mask = QU (an E-step): the one-solve form of round 6"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from hetmogp_amd.engine import Engine  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402

H_SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
C4_SPECS = [("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {}), ("Gaussian", {"sigma": 0.5}),
            ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
C5_SPECS = [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})]


def workload(name):
    """-> (specs, N_resident, M, Q, P, seed, extra kwargs of elbo_grad, engine kwargs, parameter tweak)"""
    if name == "H":
        return H_SPECS, 200000, 1024, 3, 1, 20260929, {}, {}, None
    if name == "C2":
        return H_SPECS, 200000, 512, 3, 1, 20260931, {}, {}, None
    if name in ("C3", "C3E"):
        B = 8192
        kw = dict(row_begin=[123456] * 4, row_end=[123456 + B] * 4, batch_scale=[1000000 / float(B)] * 4)
        if name == "C3E":
            kw["group_mask"] = 1
        return H_SPECS, 1000000, 1024, 3, 1, 20260932, kw, (dict(cache_kuu=True) if name == "C3E" else {}), None
    if name == "C4":
        return C4_SPECS, 125000, 1024, 4, 1, 20260933, {}, {}, None
    if name == "C5":
        return C5_SPECS, 50000, 2048, 2, 2, 20260934, {}, {}, None
    if name == "HE":
        return H_SPECS, 200000, 1024, 3, 1, 20260929, dict(group_mask=1), dict(cache_kuu=True), None
    if name == "HS":
        return H_SPECS, 200000, 1024, 3, 1, 20260929, {}, dict(strict_qf=True), None
    if name == "HSL":
        def ladder(prm, M):
            prm["lengthscale"] = np.full_like(prm["lengthscale"], 4.0 / (M - 1))
        return H_SPECS, 200000, 1024, 3, 1, 20260929, dict(forced_rung=[0, 0, 0]), dict(strict_qf=True), ladder
    if name == "HSE":
        return H_SPECS, 200000, 1024, 3, 1, 20260929, dict(group_mask=1), dict(strict_qf=True), None
    if name == "HD":
        def dense(prm, M):
            prm["lengthscale"] = np.full_like(prm["lengthscale"], 40.0 / (M - 1))
        return H_SPECS, 200000, 1024, 3, 1, 20260929, dict(forced_rung=[4, 4, 4]), {}, dense
    raise SystemExit("unknown workload " + name)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "H"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    specs, N, M, Q, P, seed, kw, ekw, tweak = workload(name)
    prm, X, Y = make_case(specs, [N] * len(specs), M=M, Q=Q, P=P, seed=seed)
    if tweak:
        tweak(prm, M)
    e = Engine(specs, Q, M, P, reuse_outputs=True, **ekw)
    e.set_data(X, Y)
    for _ in range(warm):
        out = e.elbo_grad(**dict(prm, **kw))
    t0 = time.perf_counter()
    cat = {}
    for _ in range(steps):
        out = e.elbo_grad(**dict(prm, **kw))
        for k, v in e.timings()[0].items():
            cat[k] = cat.get(k, 0.0) + v
    dt = (time.perf_counter() - t0) / max(steps, 1)
    print("%s: %.3f ms/step  ELBO %.8g rungs %s v_negative %s  %s" %
          (name, 1e3 * dt, out["elbo"], out["rungs"], out["v_negative"], {k: round(v / max(steps, 1), 3) for k, v in cat.items()}))


if __name__ == "__main__":
    main()
