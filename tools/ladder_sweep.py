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
print("%-5s %-9s %-6s | %-34s | %-34s" % ("l/h", "cond", "rung", "default: worst norm / elem-excess (array)", "strict: worst norm / elem-excess (array)"))
for c in (0.8, 1.3, 1.6, 2.0, 2.3, 2.6, 3.0, 4.0, 8.0):
    prm["lengthscale"] = np.full(Q, c * h)
    with np.errstate(all="ignore"):
        lit = so.elbo_grad_literal(prm, prob, X, Y)
    rungs = lit["rungs"]
    Kuu, Luu, Kuui, _ = so.latent_covariances(prm, prob, rungs)
    cond = max(np.linalg.cond(Luu[q] @ Luu[q].T) for q in range(Q))
    row = []
    for strict in (False, True):
        e = Engine(SPECS, Q, M, P, strict_qf=strict)
        e.set_data(X, Y)
        try:
            out = e.elbo_grad(forced_rung=rungs, **prm)
            wn = max((rel_norm(out[k], lit[k]), k) for k in KEYS)
            we = max((elementwise_excess(out[k], lit[k]), k) for k in KEYS)
            row.append("%.1e (%s) / %.2g (%s)" % (wn[0], wn[1], we[0], we[1]))
        except Exception as exc:                       # noqa: BLE001
            row.append("failed: %s" % str(exc)[:24])
        e.close()
    print("%-5.2g %-9.1e %-6s | %-34s | %-34s" % (c, cond, rungs[0], row[0], row[1]))
