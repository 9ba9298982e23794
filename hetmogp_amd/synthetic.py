"""Synthetic workloads in the style of the reference's demo (notebooks/demo.ipynb cells 1-3, hetmogp/util.py:21-50,
92-103): sorted uniform inputs per task, sinusoid-mixture latent functions mixed by W, observations drawn from each
task's likelihood; inducing inputs on a grid; lengthscale = c * (inducing spacing) so that K_uu stays well
conditioned at large M.  Used by bench.py, smoke() and the size-property tests."""
import numpy as np

_DIM_F = dict(Gaussian=1, Bernoulli=1, HetGaussian=2, Poisson=1, Exponential=1, Gamma=2, Beta=2)


def _dim_f(name, kw):
    return kw["K"] - 1 if name == "Categorical" else _DIM_F[name]


def _true_u(rng, x, Q):
    """util.true_u_functions: three-sinusoid mixtures per latent (evaluated on the first input coordinate)."""
    amp = rng.uniform(0.5, 1.5, (Q, 3))
    freq = rng.uniform(1.0, 3.0, (Q, 3))
    shift = 2.0 * rng.rand(Q, 3)
    t = x[:, :1]
    return np.hstack([3 * amp[q, 0] * np.cos(freq[q, 0] * np.pi * t + shift[q, 0] * np.pi)
                      - 2 * amp[q, 1] * np.sin(2 * freq[q, 1] * np.pi * t + shift[q, 1] * np.pi)
                      + amp[q, 2] * np.cos(4 * freq[q, 2] * np.pi * t + shift[q, 2] * np.pi) for q in range(Q)])


def _sample(rng, name, kw, F):
    """The reference's `<likelihood>.samples` link functions, on a damped F so that counts / rates stay moderate."""
    F = 0.3 * F
    n = F.shape[0]
    if name == "Gaussian":
        return F[:, :1] + kw.get("sigma", 0.5) * rng.randn(n, 1)
    if name == "HetGaussian":
        return F[:, :1] + np.exp(0.5 * F[:, 1:2]) * rng.randn(n, 1)
    if name == "Bernoulli":
        return (rng.rand(n, 1) < 1.0 / (1.0 + np.exp(-F[:, :1]))).astype(float)
    if name == "Poisson":
        return rng.poisson(np.exp(np.clip(F[:, :1], -5, 3))).astype(float)
    if name == "Exponential":
        return rng.exponential(np.exp(-np.clip(F[:, :1], -3, 3)))
    if name == "Gamma":
        return rng.gamma(np.exp(np.clip(F[:, :1], -2, 2)), 1.0 / np.exp(np.clip(F[:, 1:2], -2, 2))) + 1e-6
    if name == "Beta":
        return np.clip(rng.beta(np.exp(np.clip(F[:, :1], -2, 2)), np.exp(np.clip(F[:, 1:2], -2, 2))), 1e-6, 1 - 1e-6)
    if name == "Categorical":
        K = kw["K"]
        e = np.exp(F[:, :K - 1])
        p = np.hstack([e, np.ones((n, 1))]) / (1.0 + e.sum(1, keepdims=True))
        return (1 + (rng.rand(n, 1) > np.cumsum(p, 1)).sum(1, keepdims=True)).astype(float).clip(1, K)
    raise ValueError(name)


def make_case(specs, Ns, M, Q, P=1, seed=0, c=(0.8, 1.0, 1.3, 1.1, 0.9, 1.2, 1.0, 0.85)):
    """Returns (params dict for Engine.elbo_grad, X list, Y list)."""
    rng = np.random.RandomState(seed)
    Df = sum(_dim_f(n, k) for n, k in specs)
    X = [np.sort(rng.rand(n, P), axis=0) if P == 1 else rng.rand(n, P) for n in Ns]
    W = np.where(rng.rand(Q, Df) < 0.5, 1.0, -1.0) * rng.normal(0.5, 0.5, (Q, Df))     # util.random_W_kappas
    Y, d = [], 0
    for (name, kw), x in zip(specs, X):
        J = _dim_f(name, kw)
        Y.append(_sample(rng, name, kw, _true_u(rng, x, Q) @ W[:, d:d + J]))
        d += J
    if P == 1:
        base, h = np.linspace(0, 1, M)[:, None], 1.0 / max(M - 1, 1)
    else:
        g = int(np.ceil(M ** (1.0 / P)))
        base = np.stack(np.meshgrid(*[np.linspace(0, 1, g)] * P, indexing="ij"), -1).reshape(-1, P)[:M]
        h = 1.0 / max(g - 1, 1)
    Z = np.tile(base, (1, Q))                                                          # svmogp.py:52
    r, cc = np.tril_indices(M)
    L = [np.eye(M) + 0.05 / np.sqrt(M) * np.tril(rng.randn(M, M), -1) for _ in range(Q)]
    prm = dict(Z=Z, m_u=2.5 * rng.randn(M, Q) * 0.2, L_flat=np.stack([l[r, cc] for l in L], 1),
               variance=np.full(Q, 0.5), lengthscale=np.array([c[q % len(c)] for q in range(Q)]) * h, W=W,
               kappa=np.zeros((Q, Df)))
    return prm, X, Y
