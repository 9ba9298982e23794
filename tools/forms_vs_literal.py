#!/usr/bin/env python
"""Which strict form is closer to the REFERENCE'S OPERATIONS at large M in the jitter-ladder regime?  The literal restatement of the
reference (oracle/svmogp_oracle.py:inference_literal: LAPACK dpotrf / dpotrs / dtrmm in the reference's order; O(N^2), so a few
thousand rows) against a full-gradient strict evaluation of the engine forced into the one-solve form (HMOGP_STRICT_FORM=1) and
into the two-solve form (HMOGP_STRICT_FORM=2), lengthscale = c inducing spacings, jitter rung 0 forced on both sides.
python tools/forms_vs_literal.py [M] [ell_over_spacing] [rows_per_task]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

KEYS = ("elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z")
SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]


def case(M, c, N):
    from hetmogp_amd.synthetic import make_case
    prm, X, Y = make_case(SPECS, [N, N + 37, N - 11, N + 3], M=M, Q=2, P=1, seed=5)
    prm["lengthscale"] = np.full(2, c / (M - 1.0))
    return prm, X, Y


def child(path, M, c, N):
    from hetmogp_amd.engine import Engine
    prm, X, Y = case(M, c, N)
    e = Engine(SPECS, 2, M, 1, strict_qf=True)
    e.set_data(X, Y)
    out = e.elbo_grad(forced_rung=[0, 0], **prm)
    np.savez(path, cond=np.array(out["cond_est"]), **{k: np.asarray(out[k], float) for k in KEYS})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    c = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
    from oracle import svmogp_oracle as so
    prm, X, Y = case(M, c, N)
    prob = so.make_problem(SPECS, 2, M, 1)
    with np.errstate(all="ignore"):
        lit = so.elbo_grad_literal(prm, prob, X, Y, None, forced_rungs=[0, 0])
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, form in (("one-solve", "1"), ("two-solve", "2")):
            p = os.path.join(d, tag + ".npz")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", p, str(M), str(c), str(N)],
                                  env=dict(os.environ, HMOGP_STRICT_FORM=form))
            res[tag] = dict(np.load(p))
    print("M=%d ell/h=%.2f rows %d  cond_est %s   (worst element-wise excess over |a-b| <= 1e-5|b| + 1e-9 max|b| vs the literal restatement)" %
          (M, c, 4 * N + 29, ["%.2g" % v for v in res["one-solve"]["cond"]]))
    for k in KEYS:
        b = np.ravel(np.asarray(lit[k], float))
        row = []
        for tag in ("one-solve", "two-solve"):
            a = np.ravel(res[tag][k])
            row.append(float(np.max(np.abs(a - b) / (1e-5 * np.abs(b) + 1e-9 * np.max(np.abs(b)) + 1e-300))))
        print("  %-14s one-solve %.3g   two-solve %.3g" % (k, row[0], row[1]))
