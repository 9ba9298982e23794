"""GPU probe (round 5): how far are the device factorisation / inverse of an ill-conditioned K_uu from LAPACK's, and which of the
two carries the engine's m_fd error in the ladder regime?  Uses tests/golden/lad_h_mix_M128_ladder.npz (cond 1e7 at rung 0)."""
import os
import sys

import numpy as np
import scipy.linalg as sl
from scipy.linalg import lapack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hetmogp_amd import engine as E          # noqa: E402
from oracle import svmogp_oracle as so       # noqa: E402

for name in ("lad_h_mix_M128_ladder.npz", "lad_c1_notebook_ell.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    prm, prob, X, Y, bs = so.load_case(g)
    q, P, M = 0, prob["P"], prob["M"]
    Zq = prm["Z"][:, q * P:(q + 1) * P]
    K = so.rbf_K(Zq, Zq, prm["variance"][q], prm["lengthscale"][q])
    rung = int(g["rungs"][q])
    jit = 0.0 if rung < 0 else np.diag(K).mean() * 1e-6 * 10.0 ** rung
    Kj = K + jit * np.eye(M)
    L_ref, info = lapack.dpotrf(Kj, lower=1)
    L_ref = np.tril(L_ref)
    Ki_ref, _ = so.potri_sym(L_ref)
    L_gpu, Ki_gpu, r = E.jitchol_inv(K[None], forced_rung=[rung])
    L_gpu, Ki_gpu = L_gpu[0], Ki_gpu[0]
    nrm = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    print(name, "rung", rung, r, "| L gpu vs lapack %.2e" % nrm(L_gpu, L_ref), "| Kuui gpu vs lapack %.2e" % nrm(Ki_gpu, Ki_ref))
    # residuals: how well does each factor reproduce Kj, each inverse invert it
    print("   residual |L L^T - Kj| / |Kj|: lapack %.2e gpu %.2e" % (nrm(L_ref @ L_ref.T, Kj), nrm(L_gpu @ L_gpu.T, Kj)))
    I = np.eye(M)
    print("   |Kuui Kj - I|: lapack %.2e gpu %.2e" % (np.max(np.abs(Ki_ref @ Kj - I)), np.max(np.abs(Ki_gpu @ Kj - I))))
    # m_fd-like functional: k^T Kuui m for the fixture's rows of task 0, with each (L, inverse) combination
    Kh = so.rbf_K(X[0], Zq, prm["variance"][q], prm["lengthscale"][q])
    m = prm["m_u"][:, q]
    ref = Kh @ lapack.dpotrs(np.asfortranarray(L_ref), m, lower=1)[0]
    for tag, val in (("lapack explicit inverse", Kh @ (Ki_ref @ m)),
                     ("gpu L + substitution", Kh @ lapack.dpotrs(np.asfortranarray(L_gpu), m, lower=1)[0]),
                     ("gpu L + scipy Linv^T Linv", Kh @ ((lambda Li: Li.T @ (Li @ m))(sl.solve_triangular(L_gpu, I, lower=True)))),
                     ("gpu explicit inverse", Kh @ (Ki_gpu @ m))):
        print("   k^T Kuu^-1 m, %-28s vs dpotrs(lapack L): %.2e" % (tag, nrm(val, ref)))
