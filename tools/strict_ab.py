#!/usr/bin/env python
"""A/B of the strict q(f) mode's triangular solves: round-6 panel kernels (trsm_panel.hip) against the round-5 path
(HMOGP_TRSM_PANEL=0), same inputs, two processes.  Prints per array max|a-b| / max|b| and the worst element-wise excess over the
1e-5 criterion; also potrs_rows against scipy's cho_solve for both.
python tools/strict_ab.py [rows_per_task] [M] [Q] [ell_over_spacing] [forms]
`forms` (any 5th argument): compare the ONE-solve form forced for a full-gradient evaluation (HMOGP_STRICT_FORM=1) with the two-solve
form (HMOGP_STRICT_FORM=2) instead of the two kernel generations."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

KEYS = ("elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z")


def child(path, N, M, Q, c):
    from hetmogp_amd.engine import Engine
    from hetmogp_amd.synthetic import make_case
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
    prm, X, Y = make_case(specs, [N, N + 37, N - 11, N + 3], M=M, Q=Q, P=1, seed=5)
    if c > 0:
        prm["lengthscale"] = np.full(Q, c / (M - 1.0))
    e = Engine(specs, Q, M, 1, strict_qf=True)
    e.set_data(X, Y)
    out = e.elbo_grad(**prm)
    np.savez(path, rungs=np.array(out["rungs"]), cond=np.array(out["cond_est"]),
             **{k: np.asarray(out[k], float) for k in KEYS})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]))
        sys.exit(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    Q = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    c = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    res = {}
    forms = len(sys.argv) > 5
    with tempfile.TemporaryDirectory() as d:
        for tag, env in (("panel", "1"), ("round5", "0")):
            p = os.path.join(d, tag + ".npz")
            extra = dict(HMOGP_STRICT_FORM=("1" if tag == "panel" else "2")) if forms else dict(HMOGP_TRSM_PANEL=env)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", p, str(N), str(M), str(Q), str(c)],
                                  env=dict(os.environ, **extra))
            res[tag] = dict(np.load(p))
    if forms:
        print("(one-solve form forced vs two-solve form)")
    a, b = res["panel"], res["round5"]
    print("N=%d M=%d Q=%d ell/h=%s  rungs %s / %s  cond_est %s" % (N, M, Q, c or "default", a["rungs"], b["rungs"],
                                                                   ["%.2g" % v for v in a["cond"]]))
    for k in KEYS:
        x, y = np.ravel(a[k]), np.ravel(b[k])
        rn = np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-300)
        ex = np.max(np.abs(x - y) / (1e-5 * np.abs(y) + 1e-9 * np.max(np.abs(y)) + 1e-300))
        print("  %-14s norm %.2e   element-wise excess %.3g" % (k, rn, ex))
