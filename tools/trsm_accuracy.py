#!/usr/bin/env python
"""How accurate are the blocked triangular solves?  hmogp_potrs_rows (panel kernels for >= 1024 rows, round-5 kernels for a 333-row
slice of the same right-hand sides) and LAPACK's dpotrs against a LONG-DOUBLE substitution, on an RBF K_uu at jitter rung 0
(cond ~ 1e7) with smooth right-hand sides (rows of K_fu) and rough ones (random rows scaled like K_uu^-1 S - I).
python tools/trsm_accuracy.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sl
from hetmogp_amd.engine import potrs_rows

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = 1100
rng = np.random.RandomState(1)
h, var = 1.0 / (M - 1), 0.5
ell = 4 * h
Z = np.linspace(0, 1, M)
Xr = np.sort(rng.rand(n))
Kuu = var * np.exp(-0.5 * (Z[:, None] - Z[None, :]) ** 2 / ell ** 2) + np.eye(M) * var * 1e-6
Kfu = var * np.exp(-0.5 * (Xr[:, None] - Z[None, :]) ** 2 / ell ** 2)
rough = rng.randn(n, M) * 1e6
L = np.linalg.cholesky(Kuu)
LD = np.longdouble
Ll = L.astype(LD)


def solve_ld(B):            # rows of B: x L^T = b then a L = x, in long double, with the SAME (double) factor
    Y = B.T.astype(LD).copy()
    for j in range(M):
        Y[j] = (Y[j] - Ll[j, :j] @ Y[:j]) / Ll[j, j]
    for j in range(M - 1, -1, -1):
        Y[j] = (Y[j] - Ll[j + 1:, j] @ Y[j + 1:]) / Ll[j, j]
    return Y.T


print("M = %d, cond(K_uu) = %.2e" % (M, np.linalg.cond(Kuu)))
for name, B in (("smooth (K_fu rows)", Kfu), ("rough (1e6 N(0,1))", rough)):
    ref = solve_ld(B)
    sc = float(np.max(np.abs(ref)))
    lap = sl.cho_solve((L, True), B.T).T
    gpu = potrs_rows(L, B)                 # 1100 rows: the panel kernels (M a multiple of 128)
    old = potrs_rows(L, B[:333])           # 333 rows: the round-5 kernels
    err = lambda A, R: float(np.max(np.abs(A - R)) / sc)
    print("  %-20s max|x| %.2e   LAPACK %.2e   panel kernels %.2e   round-5 kernels %.2e   (of max|x|, vs long double)" %
          (name, sc, err(lap, ref), err(gpu, ref), err(old, ref[:333])))
