"""GPU sweep (round 5): up to which cond(K_uu) does each mode of the engine stay within the element-wise 1e-5 criterion?
Headline likelihood mix, M = 128, Q = 3, ~400 rows per task, Z = linspace, lengthscale = c x inducing spacing for a range of c.
The yardstick is the oracle's LITERAL restatement (the reference's own operations on the same LAPACK: bit-identical to the reference's
numbers on every lad_* / ref_* fixture, tests/test_oracle_golden.py); both engine modes run at the rung LAPACK takes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import elementwise_excess, rel_norm     # noqa: E402
from hetmogp_amd.engine import Engine                 # noqa: E402
from hetmogp_amd.synthetic import make_case           # noqa: E402
from oracle import svmogp_oracle as so                # noqa: E402

SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
KEYS = ["elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"]
M, Q, P = 128, 3, 1
prm, X, Y = make_case(SPECS, [400, 383, 417, 350], M=M, Q=Q, P=P, seed=20260935)
prm["Z"] = np.tile(np.linspace(0, 1, M)[:, None], (1, Q))
prob = so.make_problem(SPECS, Q, M, P)
h = 1.0 / (M - 1)
print("element-wise excess = max |a - b| / (1e-5 |b| + 1e-9 max|b|): <= 1 passes.  g_variance (a difference of sums ~1e6 x its size once")
print("cond > 1e8: the reference's own value moves by 30 % under a one-ulp change of the variance) is listed apart.")
print("%-5s %-9s %-5s | %-30s | %-30s | %-30s" % ("l/h", "cond", "rung", "reference, variance + 1 ulp", "engine default", "engine strict"))
for c in (0.8, 1.3, 1.6, 2.0, 2.3, 2.6, 3.0, 4.0, 8.0):
    prm["lengthscale"] = np.full(Q, c * h)
    with np.errstate(all="ignore"):
        lit = so.elbo_grad_literal(prm, prob, X, Y)
    rungs = lit["rungs"]
    Kuu, Luu, Kuui, _ = so.latent_covariances(prm, prob, rungs)
    cond = max(np.linalg.cond(Luu[q] @ Luu[q].T) for q in range(Q))
    def fmt(out):
        ks = [k for k in KEYS if k != "g_variance"]
        we = max((elementwise_excess(out[k], lit[k]), k) for k in ks)
        return "%.2g (%s), g_variance %.2g" % (we[0], we[1], elementwise_excess(out["g_variance"], lit["g_variance"]))
    prm2 = dict(prm, variance=prm["variance"] * (1.0 + 2.0 ** -52))
    with np.errstate(all="ignore"):
        lit2 = so.elbo_grad_literal(prm2, prob, X, Y, forced_rungs=rungs)
    row = [fmt(lit2)]
    for strict in (False, True):
        e = Engine(SPECS, Q, M, P, strict_qf=strict)
        e.set_data(X, Y)
        try:
            out = e.elbo_grad(forced_rung=rungs, **prm)
            row.append(fmt(out))
        except Exception as exc:                       # noqa: BLE001
            row.append("failed: %s" % str(exc)[:24])
        e.close()
    print("%-5.2g %-9.1e %-5s | %-30s | %-30s | %-30s" % (c, cond, rungs[0], row[0], row[1], row[2]))
