"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's variational expectations.

CPU oracle for rows L1-L8 / A6 of SURVEY.md section 8(a).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this; the product path (`hetmogp_amd/`) never does.
Pinned against `tests/golden/lik_*.npz`, which were produced by the reference's own
`likelihoods/*.py` (see `oracle/make_golden.py`).

Each function returns `(ve (N,), dm (N,dim_f), dv (N,dim_f))` where
  ve = E_q[log p(y|f)],  dm = d ve / d m,  dv = d ve / d v      (q(f) = N(m, diag v))
exactly as the reference computes them -- including its quirks:
  Q1  Gamma/Beta divide the Gauss-Hermite weights by sqrt(pi) twice (gamma.py:110,139-141,152,186-189;
      beta.py:113,142-144,155,189-192): all three outputs are 1/pi times the true 2-D quadrature.
  Q2  Categorical dm is the constant onehot(y)[d] - 1 (categorical.py:102-113).
Likelihood ids (shared with include/hetmogp_hip.h):
"""
import numpy as np
from scipy import special

LIK_GAUSSIAN, LIK_BERNOULLI, LIK_HETGAUSSIAN, LIK_CATEGORICAL, LIK_POISSON, LIK_EXPONENTIAL, LIK_GAMMA, LIK_BETA = range(8)
LIK_IDS = dict(Gaussian=LIK_GAUSSIAN, Bernoulli=LIK_BERNOULLI, HetGaussian=LIK_HETGAUSSIAN,
               Categorical=LIK_CATEGORICAL, Poisson=LIK_POISSON, Exponential=LIK_EXPONENTIAL, Gamma=LIK_GAMMA,
               Beta=LIK_BETA)

_LIM_VAL = np.log(np.finfo(np.float64).max)   # GPy safe_exp clip
_SQRT_MAX = np.sqrt(np.finfo(np.float64).max)  # GPy safe_square clip
_SQRT_PI = np.sqrt(np.pi)


def dim_f(name, K=None):
    """Number of latent parameter functions of a likelihood (`*/get_metadata`)."""
    if name == "Categorical":
        return K - 1
    return dict(Gaussian=1, Bernoulli=1, HetGaussian=2, Poisson=1, Exponential=1, Gamma=2, Beta=2)[name]


def safe_exp(f):
    return np.exp(np.minimum(f, _LIM_VAL))


def safe_square(f):
    return np.minimum(f, _SQRT_MAX) ** 2


def gh_rule(T):
    """Gauss-Hermite nodes and weights/sqrt(pi) (GPy `_gh_points` + the callers' normalisation)."""
    x, w = np.polynomial.hermite.hermgauss(T)
    return x, w / _SQRT_PI


# ------------------------------------------------------------------ closed forms (L1, L2)
def gaussian(y, m, v, sigma=0.5):
    """gaussian.py:41-62."""
    y, m, v = y.reshape(-1), m.reshape(-1), v.reshape(-1)
    s2 = sigma * sigma
    ve = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * (y * y + m * m + v - 2 * m * y) / s2
    dm = -(m - y) / s2
    dv = np.full_like(m, -0.5 / s2)
    return ve, dm[:, None], dv[:, None]


def hetgaussian(y, m, v):
    """hetgaussian.py:46-73 (function 0 = mean, function 1 = log-variance)."""
    y = y.reshape(-1)
    m1, m2, v1, v2 = m[:, 0], m[:, 1], v[:, 0], v[:, 1]
    prec = np.clip(safe_exp(-m2 + 0.5 * v2), -1e9, 1e9)
    sq = np.clip(safe_square(y) + safe_square(m1) + v1 - 2 * m1 * y, -1e9, 1e9)
    ve = -0.5 * np.log(2 * np.pi) - 0.5 * m2 - 0.5 * prec * sq
    dm = np.stack([prec * (y - m1), 0.5 * (prec * sq - 1.0)], 1)
    dv = np.stack([-0.5 * prec, -0.25 * prec * sq], 1)
    return ve, dm, dv


# ------------------------------------------------------------------ 1-D quadrature (L3-L5)
def _quad1d(y, m, v, fns, T=20):
    y, m, v = y.reshape(-1), m.reshape(-1), v.reshape(-1)
    x, w = gh_rule(T)
    f = x[None, :] * np.sqrt(2.0 * v[:, None]) + m[:, None]
    logp, d1, d2 = fns(f, y[:, None])
    return logp @ w, (d1 @ w)[:, None], (0.5 * (d2 @ w))[:, None]


def bernoulli(y, m, v):
    """bernoulli.py:31-36,66-111."""
    def fns(f, yy):
        ef = safe_exp(f)
        p = np.clip(ef / (1 + ef), 1e-9, 1 - 1e-9)
        logp = yy * np.log(p) + (1 - yy) * np.log(1 - p)
        d1 = ((yy - p) / (1 - p)) * (1 / (1 + ef))
        d2 = -p / (1 + ef)
        return logp, d1, d2
    return _quad1d(y, m, v, fns)


def poisson(y, m, v):
    """poisson.py:31-34,56-95."""
    def fns(f, yy):
        ef = safe_exp(f)
        return -ef + yy * f - special.gammaln(yy + 1), yy - ef, -ef
    return _quad1d(y, m, v, fns)


def exponential(y, m, v):
    """exponential.py:28-32,58-99."""
    def fns(f, yy):
        b = np.clip(safe_exp(-f), 1e-9, 1e9)
        return -np.log(b) - yy / b, 1 - yy / b, -yy / b
    return _quad1d(y, m, v, fns)


# ------------------------------------------------------------------ 2-D quadrature (L6, L7)
def _grid2(m, v, T=10):
    x, w = gh_rule(T)
    f1 = x[None, :] * np.sqrt(2.0 * v[:, 0, None]) + m[:, 0, None]          # (N,T)
    f2 = x[None, :] * np.sqrt(2.0 * v[:, 1, None]) + m[:, 1, None]
    return f1[:, :, None], f2[:, None, :], w


def _contract2(g, w):
    # reference: g.dot(gh_w)/sqrt(pi) over the last axis, then again -- with gh_w ALREADY /sqrt(pi) (quirk Q1)
    return ((g @ w) / _SQRT_PI) @ w / _SQRT_PI


def gamma(y, m, v):
    """gamma.py:34-41,80-194: a = exp(f1) (shape), b = exp(f2) (rate), both clipped to [1e-9,1e9]."""
    y = y.reshape(-1)[:, None, None]
    f1, f2, w = _grid2(m, v)
    a = np.clip(safe_exp(f1), 1e-9, 1e9) + 0 * f2
    b = np.clip(safe_exp(f2), 1e-9, 1e9) + 0 * f1
    logy = np.log(y)
    psi_a = special.psi(a)
    logp = -special.gammaln(a) + a * np.log(b) + (a - 1) * logy - b * y
    d1a = (-psi_a + np.log(b) + logy) * a
    d1b = a - b * y
    d2a = (-psi_a - a * special.zeta(2, a) + np.log(b) + logy) * a
    d2b = -y * b
    ve = _contract2(logp, w)
    dm = np.stack([_contract2(d1a, w), _contract2(d1b, w)], 1)
    dv = 0.5 * np.stack([_contract2(d2a, w), _contract2(d2b, w)], 1)
    return ve, dm, dv


def beta(y, m, v):
    """beta.py:29-36,76-197."""
    y = y.reshape(-1)[:, None, None]
    f1, f2, w = _grid2(m, v)
    a = np.clip(safe_exp(f1), 1e-9, 1e9) + 0 * f2
    b = np.clip(safe_exp(f2), 1e-9, 1e9) + 0 * f1
    logy, log1y = np.log(y), np.log(1 - y)
    psi_ab, psi_a, psi_b = special.psi(a + b), special.psi(a), special.psi(b)
    z_ab, z_a, z_b = special.zeta(2, a + b), special.zeta(2, a), special.zeta(2, b)
    logp = (a - 1) * logy + (b - 1) * log1y - special.betaln(a, b)
    d1a = (psi_ab - psi_a + logy) * a
    d1b = (psi_ab - psi_b + log1y) * b
    d2a = (psi_ab + a * z_ab - psi_a - a * z_a + logy) * a
    d2b = (psi_ab + b * z_ab - psi_b - b * z_b + log1y) * b
    ve = _contract2(logp, w)
    dm = np.stack([_contract2(d1a, w), _contract2(d1b, w)], 1)
    dv = 0.5 * np.stack([_contract2(d2a, w), _contract2(d2b, w)], 1)
    return ve, dm, dv


# ------------------------------------------------------------------ (K-1)-D quadrature (L8)
def categorical(y, m, v, K, T=10, chunk=256, exact_dm=False):
    """categorical.py:37-46,77-82,102-222.  Labels are 1..K (quirk Q7); class K is the reference class."""
    N, D = m.shape
    assert D == K - 1
    y = y.reshape(-1)
    x, w = gh_rule(T)
    wsum = np.sum(np.polynomial.hermite.hermgauss(T)[1]) / _SQRT_PI
    ve = np.empty(N)
    dm = np.empty((N, D))
    dv = np.empty((N, D))
    onehot = (y[:, None] == (np.arange(K)[None, :] + 1)).astype(float)      # (N,K)
    valid = onehot.sum(1)
    # tensor weights over the D-dim grid
    W = w
    for _ in range(D - 1):
        W = np.multiply.outer(W, w)
    Wf = W.reshape(-1)
    grids = np.stack(np.meshgrid(*[x] * D, indexing="ij"), -1).reshape(-1, D)    # (T^D, D)
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        F = grids[None, :, :] * np.sqrt(2.0 * v[s:e, None, :]) + m[s:e, None, :]  # (n,G,D)
        eF = safe_exp(F)
        den = 1.0 + eF.sum(-1, keepdims=True)
        p = np.concatenate([eF / den, 1.0 / den], -1)
        p = np.clip(p, 1e-9, 1 - 1e-9)
        p = p / p.sum(-1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            logp = np.sum(special.xlogy(onehot[s:e, None, :], p), -1)            # multinomial.logpmf, n=1
        logp = np.where(valid[s:e, None] == 1.0, logp, np.nan)
        ve[s:e] = logp @ Wf
        for d in range(D):
            enum = safe_exp(F + F[:, :, d:d + 1])
            enum[:, :, d] = safe_exp(F[:, :, d])
            pd = enum.sum(-1) / safe_square(den[..., 0])
            d2 = -valid[s:e, None] * pd
            dv[s:e, d] = 0.5 * (d2 @ Wf)
            if exact_dm:    # true derivative E[onehot_d - softmax_d] (quirks = "exact"; not the reference)
                dm[s:e, d] = (valid[s:e, None] * (onehot[s:e, None, d] - eF[:, :, d] / den[..., 0])) @ Wf
    if exact_dm:
        return ve, dm, dv
    # quirk Q2: p/p == 1 -> dlogp = onehot_d - sum_k onehot_k, integrated against weights summing to ~1
    dm[:, :] = (onehot[:, :D] - valid[:, None]) * (wsum ** D)
    return ve, dm, dv


def var_exp_all(name, y, m, v, exact=False, **kw):
    """Dispatch on the reference's class name (het_likelihood.py:101-131 loops these per task).  exact=True is NOT the
    reference: it removes quirk Q1 (the extra 1/pi of Gamma / Beta) and quirk Q2 (Categorical's constant d/dm) so that
    (dm, dv) are the true derivatives of ve -- the yardstick of the engine's quirks = "exact" mode."""
    if name == "Gaussian":
        return gaussian(y, m, v, kw.get("sigma", 0.5) if kw.get("sigma", None) is not None else 0.5)
    if name == "Categorical":
        return categorical(y, m, v, kw["K"], exact_dm=exact)
    out = dict(Bernoulli=bernoulli, HetGaussian=hetgaussian, Poisson=poisson, Exponential=exponential, Gamma=gamma,
               Beta=beta)[name](y, m, v)
    if exact and name in ("Gamma", "Beta"):
        out = tuple(np.pi * o for o in out)
    return out


# ============================================================================ predictive
def predictive(name, m, v, gh_T=None, **kw):
    """`<likelihood>.predictive(m, v)` of the reference: predictive mean and variance of y under q(f) = N(m, diag v).
    Returns (mean_pred (N, dim_p), var_pred (N, dim_p)).  gh_T = Gauss-Hermite order the instance would use: 20 on a
    fresh instance, 10 for Gamma/Beta whose var_exp ran first (quirk Q7); Categorical always 10.
    References: gaussian.py:64-67, bernoulli.py:113-128, hetgaussian.py:75-88, poisson.py:97-112, exponential.py:101-116,
    gamma.py:196-238, beta.py:199-241 (both 1/pi-scaled, quirk Q1), categorical.py:224-269 (variance 'NOT IMPLEMENTED')."""
    N = m.shape[0]
    if name == "Gaussian":
        s = kw.get("sigma", 0.5)
        s = 0.5 if s is None else s
        return m.reshape(N, 1).copy(), s * s + v.reshape(N, 1)
    if name == "HetGaussian":
        x, w = gh_rule(gh_T or 20)
        f1 = x[None, :] * np.sqrt(2.0 * v[:, 0, None]) + m[:, 0, None]
        f2 = x[None, :] * np.sqrt(2.0 * v[:, 1, None]) + m[:, 1, None]
        return m[:, :1].copy(), (safe_exp(f2) @ w + safe_square(f1) @ w - np.square(m[:, 0]))[:, None]
    if name in ("Bernoulli", "Poisson", "Exponential"):
        x, w = gh_rule(gh_T or 20)
        f = x[None, :] * np.sqrt(2.0 * v.reshape(-1)[:, None]) + m.reshape(-1)[:, None]
        if name == "Bernoulli":
            ef = safe_exp(f)
            p = np.clip(ef / (1 + ef), 1e-9, 1 - 1e-9)
            mean, var, msq = p, p * (1 - p), np.square(p)
        elif name == "Poisson":
            ef = safe_exp(f)
            mean, var, msq = ef, ef, np.square(ef)
        else:
            b = np.clip(safe_exp(-f), 1e-9, 1e9)
            mean, var, msq = b, safe_square(b), safe_square(b)
        mp = mean @ w
        return mp[:, None], (var @ w + msq @ w - np.square(mp))[:, None]
    if name in ("Gamma", "Beta"):
        f1, f2, w = _grid2(m, v, gh_T or 20)
        a = np.clip(safe_exp(f1), 1e-9, 1e9) + 0 * f2
        b = np.clip(safe_exp(f2), 1e-9, 1e9) + 0 * f1
        if name == "Gamma":
            mean, var = a / b, a / b ** 2
        else:
            mean, var = a / (a + b), a * b / ((a + b) ** 2 * (a + b + 1))
        c2 = lambda g: (g @ w) @ w / np.square(_SQRT_PI)
        mp = c2(mean)
        return mp[:, None], (c2(var) + c2(np.square(mean)) - safe_square(mp))[:, None]
    if name == "Categorical":
        K = kw["K"]
        D = K - 1
        x, w = gh_rule(10)
        Wt = w
        for _ in range(D - 1):
            Wt = np.multiply.outer(Wt, w)
        Wf = Wt.reshape(-1)
        grids = np.stack(np.meshgrid(*[x] * D, indexing="ij"), -1).reshape(-1, D)
        F = grids[None, :, :] * np.sqrt(2.0 * v[:, None, :]) + m[:, None, :]
        eF = safe_exp(F)
        rho = np.clip(eF / (1.0 + eF.sum(-1, keepdims=True)), 1e-9, 1 - 1e-9)
        rho = rho / rho.sum(-1, keepdims=True)                      # normalised over the K-1 columns (categorical.py:89-91)
        return np.einsum("ngd,g->nd", rho, Wf), np.zeros((N, D))
    raise ValueError(name)
