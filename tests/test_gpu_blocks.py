"""GPU: parity of the building-block kernels, called through the C ABI (hetmogp_amd.engine -> libhetmogp_hip.so),
against the NumPy oracle / LAPACK and the golden vectors of the reference.  Tolerances: fp64, relative to the
largest magnitude of each array (1e-12 for GEMM / RBF, 1e-9 for factorisations and quadratures)."""
import glob
import json
import os

import numpy as np
import pytest
import scipy.linalg

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


@pytest.fixture(scope="module")
def E():
    from hetmogp_amd import engine
    return engine


@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("shape", [(128, 128, 64), (130, 70, 37), (5, 5, 5), (300, 257, 129), (1, 1, 1), (64, 512, 1030),
                                   (257, 3, 16)])
def test_gemm_f64_all_layouts(E, tA, tB, shape):
    """Asymmetric operands so a transposed fragment map cannot pass (cdna guide rule 16)."""
    M, N, K = shape
    rng = np.random.RandomState(M * 7 + N * 3 + K + tA * 2 + tB)
    A = rng.randn(K, M) if tA else rng.randn(M, K)
    B = rng.randn(N, K) if tB else rng.randn(K, N)
    A += np.arange(A.shape[1])[None, :] * 0.01
    C0 = rng.randn(M, N)
    want = 0.7 * (A.T if tA else A) @ (B.T if tB else B) - 1.3 * C0
    got = E.gemm(A, B, transA=bool(tA), transB=bool(tB), alpha=0.7, beta=-1.3, C0=C0)
    assert rel(got, want) < 1e-13


@pytest.mark.parametrize("P,N,M", [(1, 100, 37), (1, 257, 64), (2, 65, 50), (3, 33, 16), (1, 5, 1030)])
def test_rbf_cross_cov(E, P, N, M):
    from oracle import svmogp_oracle as so
    rng = np.random.RandomState(P * 100 + N + M)
    X, Z = rng.rand(N, P), rng.rand(M, P)
    ell = 0.9 * M ** (-1.0 / P)
    got = E.rbf_cross_cov(X, Z, 0.7, ell)
    want = so.rbf_K(X, Z, 0.7, ell)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "cov_*.npz"))), ids=os.path.basename)
def test_jitchol_ladder_and_inverse_golden(E, path):
    """Golden Kuu -> (Luu, Kuui, rung) from the reference's util.latent_funs_cov (util.py:181-200)."""
    g = np.load(path)
    L, Ai, rungs = E.jitchol_inv(g["Kuu"])
    assert rungs == [int(r) for r in g["rung"]]
    if max(rungs) < 0:     # well-conditioned: factor and inverse agree tightly
        assert rel(L, g["Luu"]) < 1e-10
        assert rel(Ai, g["Kuui"]) < 1e-8
    else:                  # singular + jitter: cond ~ 1e6, compare through the defining identities instead
        for q in range(L.shape[0]):
            jit = np.diag(g["Kuu"][q]).mean() * 1e-6 * 10.0 ** rungs[q]
            Aj = g["Kuu"][q] + jit * np.eye(L.shape[1])
            assert rel(L[q] @ L[q].T, Aj) < 1e-12
            assert rel(Ai[q] @ Aj, np.eye(L.shape[1])) < 1e-6


@pytest.mark.parametrize("M", [1, 31, 32, 33, 50, 64, 200, 513, 1024])
def test_potrf_potri_vs_lapack(E, M):
    rng = np.random.RandomState(M)
    Q = 2
    B = rng.randn(Q, M, M)
    A = B @ B.transpose(0, 2, 1) + M * np.eye(M)[None]
    L, Ai, rungs = E.jitchol_inv(A)
    assert rungs == [-1] * Q
    for q in range(Q):
        Lr = scipy.linalg.cholesky(A[q], lower=True)
        assert rel(L[q], Lr) < 1e-12
        assert np.all(np.triu(L[q], 1) == 0.0)
        assert rel(Ai[q], np.linalg.inv(A[q])) < 1e-10
    Lv = np.tril(rng.randn(Q, M, M)) * 0.1 + np.eye(M)[None]
    Si = E.potri(Lv)
    for q in range(Q):
        assert rel(Si[q], np.linalg.inv(Lv[q] @ Lv[q].T)) < 1e-9


def test_jitchol_not_pd_raises(E):
    A = -np.eye(8)[None]
    with pytest.raises(np.linalg.LinAlgError):
        E.jitchol_inv(A)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "lik_*.npz"))), ids=os.path.basename)
def test_var_exp_golden(E, path):
    """Per-likelihood E_q[log p], d/dm, d/dv against the reference's own likelihoods/*.py outputs (incl. clip rows)."""
    g = np.load(path)
    name, kw = json.loads(str(g["spec"]))
    ve, dm, dv = E.var_exp(name, g["y"], g["m"], g["v"], **kw)
    for got, want in ((ve[:, None], g["var_exp"]), (dm, g["var_exp_dm"]), (dv, g["var_exp_dv"])):
        np.testing.assert_allclose(got, want, rtol=2e-9, atol=1e-11 * np.max(np.abs(want)))


def test_var_exp_large_random_vs_oracle(E):
    from oracle import likelihoods_oracle as lo
    rng = np.random.RandomState(5)
    n = 3000
    for name, kw, y in (("Bernoulli", {}, (rng.rand(n) < 0.4).astype(float)), ("Poisson", {}, rng.poisson(4.0, n).astype(float)),
                        ("Gamma", {}, rng.gamma(2.0, 1.0, n) + 1e-3), ("Beta", {}, np.clip(rng.beta(2, 3, n), 1e-4, 1 - 1e-4)),
                        ("Categorical", {"K": 4}, rng.randint(1, 5, n).astype(float)), ("HetGaussian", {}, rng.randn(n))):
        J = E.lik_dim_f(name, **kw)
        m, v = rng.uniform(-2, 2, (n, J)), np.exp(rng.uniform(-5, 1, (n, J)))
        got = E.var_exp(name, y, m, v, **kw)
        want = lo.var_exp_all(name, y[:, None], m, v, **kw)
        for a, b in zip(got, want):
            np.testing.assert_allclose(a, b.reshape(a.shape), rtol=1e-9, atol=1e-11 * np.max(np.abs(b)))
