"""GPU probe (round 5): are the loosened 2-D / M = 2048 tolerances (2e-7, 1e-7) the explicit-inverse FORM or the
hardware?  For 2-D inducing grids (cond(K_uu) 1e5 .. 1e6, no jitter) and the C5 shape: array-normalised distance of
  default engine vs the oracle's fused restatement (same algebra: explicit C_q)      -- what the tests assert
  default engine vs the oracle's literal restatement (the reference's solve-based forms)
  STRICT  engine vs the oracle's literal restatement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import elementwise_excess, rel_norm     # noqa: E402
from test_gpu_engine import KEYS, synth               # noqa: E402
from hetmogp_amd.engine import Engine                 # noqa: E402
from oracle import svmogp_oracle as so                # noqa: E402

CASES = [("2-D M=144", [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], [420, 500], 144, 2, 2, (0.9, 1.2)),
         ("2-D M=400", [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], [700, 900], 400, 2, 2, (0.9, 1.2)),
         ("2-D M=1024", [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], [900, 1100], 1024, 2, 2, (0.9, 1.2)),
         ("C5 shape 2-D M=2048", [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})], [1500, 1500], 2048, 2, 2, (0.9, 1.2)),
         ("1-D M=2048", [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {})], [1000, 1000, 1000], 2048, 3, 1, (0.8, 1.0, 1.3))]
print("%-22s %-9s | %-28s | %-28s | %-28s" % ("case", "cond", "default vs fused", "default vs literal", "STRICT vs literal"))
for tag, specs, Ns, M, Q, P, cs in CASES:
    prm, prob, X, Y = synth(4242, specs, Ns, M, Q, P, cs)
    fused = so.elbo_grad_fused(prm, prob, X, Y)
    lit = so.elbo_grad_literal(prm, prob, X, Y)
    Kuu, Luu, Kuui, _ = so.latent_covariances(prm, prob, lit["rungs"])
    cond = max(np.linalg.cond(Kuu[q]) for q in range(Q))
    cells = []
    for strict, ref in ((False, fused), (False, lit), (True, lit)):
        e = Engine(specs, Q, M, P, strict_qf=strict)
        e.set_data(X, Y)
        out = e.elbo_grad(**prm)
        wn = max((rel_norm(out[k], ref[k]), k) for k in KEYS)
        we = max((elementwise_excess(out[k], ref[k]), k) for k in KEYS)
        cells.append("%.1e (%s) / %.2g" % (wn[0], wn[1], we[0]))
        e.close()
    print("%-22s %-9.1e | %-28s | %-28s | %-28s" % (tag, cond, cells[0], cells[1], cells[2]))
print("(cells: worst array-normalised error (array) / worst element-wise excess over ELBO + the 7 gradient arrays)")
