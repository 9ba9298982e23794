"""Round-5 probe, CPU part: how many int8 slice products does an error-bounded split of the forward contraction
P~ = K^ C_q need?  K^ in [0, sigma^2] is sliced per ROW (7 magnitude bits per signed int8 slice), C_q per COLUMN; the product keeps the
slice pairs (a, b) with a + b < S ("S diagonals": S (S + 1) / 2 int8 GEMMs with exact int32 accumulation, recombined in FP64).
Printed per S: max |error| relative to max |P~| and relative to the row-wise scale, for the headline operand (H: lengthscale ~ one
inducing spacing) and the dense one (HD: 40 spacings, jitter rung 4) -- and the c = rowsum(P~ .* K^) statistic, which is what the ELBO sees."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hetmogp_amd.synthetic import make_case          # noqa: E402
from oracle import svmogp_oracle as so               # noqa: E402

SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
M, Q, rows = 1024, 3, 768


def slices(A, axis, S):
    """A = scale * sum_i sl[i] * 2^(-7 (i + 1)) with integer slices |sl[i]| <= 127 (sign carried by every slice)."""
    scale = np.max(np.abs(A), axis=axis, keepdims=True)
    scale = np.where(scale > 0, 2.0 ** np.ceil(np.log2(scale)), 1.0)
    R = A / scale                          # |R| <= 1
    out = []
    for _ in range(S):
        R = R * 128.0
        s = np.trunc(R)
        s = np.clip(s, -127, 127)
        out.append(s)                      # (float64 holding small integers: BLAS products of them are exact, sums < 2^53)
        R = R - s
    return scale, out


for tag, ell_c, rung in (("H  (l ~ 1 spacing)", None, None), ("HD (l = 40 spacings, rung 4)", 40.0, 4)):
    prm, X, Y = make_case(SPECS, [20000] * 4, M=M, Q=Q, P=1, seed=20260929)
    if ell_c:
        prm["lengthscale"] = np.full(Q, ell_c / (M - 1))
    prob = so.make_problem(SPECS, Q, M, 1)
    u = so.u_algebra(prm, prob, [rung] * Q if rung is not None else None)
    q = 0
    Xs = X[0][10000:10000 + rows]
    K = so.rbf_K(Xs, prm["Z"][:, q:q + 1], prm["variance"][q], prm["lengthscale"][q])
    C = u["C"][q]
    P = K @ C
    c = np.sum(P * K, 1)
    print("%s: max|K^| %.2g, max|C| %.2g, max|P~| %.2g, max|c| %.2g, cond(K_uu + jitter) %.1e" % (
        tag, K.max(), np.abs(C).max(), np.abs(P).max(), np.abs(c).max(), np.linalg.cond(u["Luu"][q] @ u["Luu"][q].T)))
    Smax = 9
    sk, Ks = slices(K, 1, Smax)
    sc, Cs = slices(C, 0, Smax)
    for S in range(3, Smax + 1):
        acc = np.zeros_like(P)
        for a in range(S):
            for b in range(S - a):
                acc += (Ks[a] @ Cs[b]) * 2.0 ** (-7 * (a + b + 2))
        Ph = acc * sk * sc
        ch = np.sum(Ph * K, 1)
        print("   S = %d diagonals = %2d int8 products: max|dP~|/max|P~| %.1e   max|dP~| / row scale %.1e   max|dc|/max|c| %.1e" % (
            S, S * (S + 1) // 2, np.max(np.abs(Ph - P)) / np.abs(P).max(),
            np.max(np.abs(Ph - P) / (np.max(np.abs(P), 1, keepdims=True) + 1e-300)), np.max(np.abs(ch - c)) / np.abs(c).max()))
