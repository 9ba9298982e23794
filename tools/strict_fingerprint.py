#!/usr/bin/env python
"""Bit-level fingerprint of one strict-mode evaluation (A/B runs of kernel variants must print the same value):
python tools/strict_fingerprint.py [rows_per_task] [M] [Q] [strict = 1]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from hetmogp_amd.engine import Engine  # noqa: E402
from hetmogp_amd.synthetic import make_case  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 2
STRICT = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
prm, X, Y = make_case(specs, [N, N + 37, N - 11, N + 3], M=M, Q=Q, P=1, seed=5)
e = Engine(specs, Q, M, 1, strict_qf=STRICT)
e.set_data(X, Y)
out = e.elbo_grad(**prm)
h = hashlib.sha256()
for k in ("elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"):
    h.update(np.ascontiguousarray(np.asarray(out[k], float)).tobytes())
print("strict=%d " % STRICT + "N=%d M=%d Q=%d elbo %.12g fingerprint %s" % (N, M, Q, out["elbo"], h.hexdigest()[:16]))
