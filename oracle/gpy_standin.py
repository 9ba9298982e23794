"""TEST INFRASTRUCTURE ONLY -- stand-in for the third-party names the reference imports.

The reference (`/root/reference`, pmorenoz/HetMOGP) depends on GPy==1.9.5, paramz and
climin 0.1a1 (`requirements.txt:2-6`), none of which is vendored or installed here.
`install()` injects minimal `sys.modules` entries so that the reference's OWN
`hetmogp/svmogp_inf.py`, `hetmogp/util.py`, `hetmogp/het_likelihood.py` and
`likelihoods/*.py` import and run unmodified.  It is used by `oracle/make_golden.py`
(in the authoring container, where `/root/reference` exists) to capture golden
vectors; it is inert on the GPU box.

The semantics restated here are those of GPy 1.9.5:
  * `GPy.util.linalg.jitchol/dpotri/dpotrs`  -- LAPACK wrappers + the jitter ladder
  * `GPy.util.choleskies.flat_to_triang/triang_to_flat` -- row-major tril packing
  * `GPy.util.misc.safe_exp/safe_square`
  * `GPy.likelihoods.Likelihood._gh_points` -- cached hermgauss
  * `GPy.kern.RBF.K/Kdiag`, `GPy.kern.Coregionalize.W/kappa/B`
  * a passive `Posterior` container
Nothing in the product path (`hetmogp_amd/`) may import this module.
"""
import sys
import types

import numpy as np
import scipy.linalg
from scipy.linalg import lapack

_LIM_VAL = np.log(np.finfo(np.float64).max)
_SQRT_MAX = np.sqrt(np.finfo(np.float64).max)


# ----------------------------------------------------------------------------- linalg
def jitchol(A, maxtries=5, _record=None):
    """GPy.util.linalg.jitchol: plain dpotrf first, then mean(diag)*1e-6*10^k, k=0..4."""
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        if _record is not None:
            _record.append(-1)
        return L
    diagA = np.diag(A)
    if np.any(diagA <= 0.0):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        try:
            L = scipy.linalg.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
            if _record is not None:
                _record.append(num_tries - 1)
            return L
        except Exception:
            jitter *= 10
        finally:
            num_tries += 1
    raise np.linalg.LinAlgError("not positive definite, even with jitter.")


def _symmetrify_lower(A):
    i, j = np.triu_indices(A.shape[0], 1)
    A[i, j] = A[j, i]


def dpotri(A, lower=1):
    A = np.asfortranarray(A)
    R, info = lapack.dpotri(A, lower=lower)
    R = np.array(R)
    _symmetrify_lower(R)
    return R, info


def dpotrs(A, B, lower=1):
    A = np.asfortranarray(A)
    return lapack.dpotrs(A, B, lower=lower)


# ------------------------------------------------------------------------- choleskies
def flat_to_triang(flat):
    N, D = flat.shape
    M = int((-1 + np.sqrt(8 * N + 1)) // 2)
    ret = np.zeros((D, M, M))
    r, c = np.tril_indices(M)
    for d in range(D):
        ret[d, r, c] = flat[:, d]
    return ret


def triang_to_flat(L):
    D, M, _ = L.shape
    r, c = np.tril_indices(M)
    flat = np.empty((M * (M + 1) // 2, D))
    for d in range(D):
        flat[:, d] = L[d, r, c]
    return flat


# ------------------------------------------------------------------------------- misc
def safe_exp(f):
    clip_f = np.clip(f, -np.inf, _LIM_VAL)
    return np.exp(clip_f)


def safe_square(f):
    f = np.clip(f, -np.inf, _SQRT_MAX)
    return f ** 2


# ------------------------------------------------------------------------ likelihoods
class Identity(object):
    def transf(self, f):
        return f


class Likelihood(object):
    def __init__(self, gp_link=None, name="likelihood"):
        self.gp_link = gp_link
        self.name = name
        self.__gh_points = None

    def _gh_points(self, T=20):
        if self.__gh_points is None:
            self.__gh_points = np.polynomial.hermite.hermgauss(T)
        return self.__gh_points



# ------------------------------------------------------------------------------ param
class Param(np.ndarray):
    """Minimal paramz.Param: an ndarray with a `.gradient` of the same shape; basic slices
    share the parent's gradient storage (so `p[:, q:q+1].gradient = g` writes through,
    as used at svmogp.py:106-113)."""

    def __new__(cls, name, input_array, *a, **kw):
        obj = np.array(np.atleast_1d(input_array), dtype=float).view(cls)   # copy == detach
        obj.name = name
        obj._grad = np.zeros(obj.shape)
        obj._fixed = [False]
        return obj

    def __array_finalize__(self, obj):
        self.name = getattr(obj, "name", None)
        self._grad = None
        self._fixed = getattr(obj, "_fixed", [False])

    def __getitem__(self, idx):
        out = np.ndarray.__getitem__(self, idx)
        if isinstance(out, Param) and getattr(self, "_grad", None) is not None:
            try:
                out._grad = self._grad[idx]
            except Exception:
                out._grad = None
        return out

    @property
    def gradient(self):
        return self._grad

    @gradient.setter
    def gradient(self, val):
        if self._grad is None:
            self._grad = np.zeros(self.shape)
        self._grad[...] = np.asarray(val).reshape(self._grad.shape)

    @property
    def values(self):
        return np.asarray(self)

    @property
    def is_fixed(self):
        return self._fixed[0]

    def fix(self):
        self._fixed[0] = True

    def unfix(self):
        self._fixed[0] = False


# paramz's Logexp transformation (transformations.py; SURVEY appendix A, restated): positive parameters are optimised through
# theta = log(1 + e^x); _lim_val = 36, the commented-out epsilon of the original is not added
_LOGEXP_LIM = 36.0


def logexp_f(x):
    return np.where(x > _LOGEXP_LIM, x, np.log1p(np.exp(np.clip(x, -_LIM_VAL, _LOGEXP_LIM))))


def logexp_finv(f):
    return np.where(f > _LOGEXP_LIM, f, np.log(np.expm1(f)))


def logexp_gradfactor(f):
    return np.where(f > _LOGEXP_LIM, 1.0, -np.expm1(-f))


class _Selection(object):
    """What `model['<regexp>']` returns as far as the reference uses it (util.py:285-315: `.fix()` / `.unfix()`)."""

    def __init__(self, params):
        self.params = params

    def fix(self):
        [p.fix() for p in self.params]

    def unfix(self):
        [p.unfix() for p in self.params]


class SparseGP(object):
    """Stand-in for GPy.core.SparseGP as used by svmogp.py:56-58 (dummy X/Y, Z as a Param) and -- [r6] -- for the paramz surface the
    reference's SVI driver touches (svmogp.py:188-199 `self._grads(parameters)`, util.py:285-329 `model['.*.kappa'].fix()`,
    `model.optimizer_array`): parameter order = link order (svmogp.py:71-75: Z at index 0, m_u, L_u, kernels, B_q), fixed parameters
    leave the flat vector, positive ones (RBF variance / lengthscale, Coregionalize kappa) go through Logexp, `_grads(x)` sets the
    vector (firing parameters_changed) and returns the gradient of the objective -log_likelihood.  SURVEY appendix A, last row."""

    def __init__(self, X, Y, Z, kernel, likelihood, mean_function=None, X_variance=None,
                 inference_method=None, name="sparse gp", Y_metadata=None, normalizer=False):
        self.X, self.Y = X, Y
        self.Z = Param("inducing inputs", Z)
        self.kern = kernel
        self.likelihood = likelihood
        self.inference_method = inference_method
        self.Y_metadata = Y_metadata
        self.name = name
        self._linked = []

    def link_parameter(self, p, index=None):
        if index is None:
            self._linked.append(p)
        else:
            self._linked.insert(index, p)

    def _leaves(self):
        """[(hierarchy name without the model's, Param, positive?)] in link order."""
        out, seen = [], {}
        for p in self._linked:
            if isinstance(p, Param):
                out.append((p.name, p, False))
                continue
            n = seen.get(p.name, 0)                     # paramz disambiguates equal names: rbf, rbf_1, ...
            seen[p.name] = n + 1
            base = p.name if n == 0 else "%s_%d" % (p.name, n)
            if isinstance(p, RBF):
                out += [(base + ".variance", p.variance, True), (base + ".lengthscale", p.lengthscale, True)]
            else:
                out += [(base + ".W", p.W, False), (base + ".kappa", p.kappa, True)]
        return out

    def __getitem__(self, pattern):
        import re
        rx = re.compile(pattern)
        hit = [p for name, p, _ in self._leaves() if rx.match(name)]
        if not hit:
            raise AttributeError(pattern)
        return _Selection(hit)

    @property
    def optimizer_array(self):
        return np.concatenate([logexp_finv(np.ravel(p)) if pos else np.ravel(np.asarray(p)).copy()
                               for _, p, pos in self._leaves() if not p.is_fixed])

    @optimizer_array.setter
    def optimizer_array(self, x):
        i = 0
        for _, p, pos in self._leaves():
            if p.is_fixed:
                continue
            v = np.asarray(x[i:i + p.size], dtype=float).reshape(p.shape)
            np.asarray(p)[...] = logexp_f(v) if pos else v
            i += p.size
        self.parameters_changed()

    def _grads(self, x):
        self.optimizer_array = x
        return -np.concatenate([np.ravel(p.gradient) * (logexp_gradfactor(np.ravel(np.asarray(p))) if pos else 1.0)
                                for _, p, pos in self._leaves() if not p.is_fixed])

    def link_parameters(self, *ps):
        self._linked.extend(ps)

    def unlink_parameter(self, p):
        pass

# ------------------------------------------------------------------------------ kerns
class RBF(object):
    def __init__(self, input_dim, variance=1.0, lengthscale=None, ARD=False, name="rbf"):
        self.input_dim = input_dim
        self.variance = Param("variance", np.atleast_1d(np.asarray(variance, dtype=float)))
        self.lengthscale = Param("lengthscale", np.atleast_1d(np.asarray(1.0 if lengthscale is None else lengthscale, dtype=float)))
        self.name = name

    def copy(self):
        return RBF(self.input_dim, self.variance.copy(), self.lengthscale.copy(), name=self.name)

    def prod(self, other, name=None):
        return _Prod()

    def _unscaled_dist(self, X, X2):
        X1sq = np.sum(np.square(X), 1)
        X2sq = np.sum(np.square(X2), 1)
        r2 = -2.0 * np.dot(X, X2.T) + (X1sq[:, None] + X2sq[None, :])
        r2 = np.clip(r2, 0, np.inf)
        return np.sqrt(r2)

    def _slice(self, X):
        """GPy's KernCallsViaSlicerMeta: every public kernel call sees X[:, active_dims] with active_dims =
        arange(input_dim) (Kern.__init__ default).  Only matters where the reference hands a kernel MORE columns than
        input_dim: `kern.K(self.Z, Xnew)` at svmogp.py:240 passes the whole M x (Q*P) inducing array, of which GPy keeps
        the first P columns (the block of latent 0) whatever q is.  Restated from GPy 1.9.5, not in the tree."""
        X = np.asarray(X)
        return X[:, :self.input_dim] if X.ndim == 2 and X.shape[1] > self.input_dim else X

    def K(self, X, X2=None):
        X = self._slice(X)
        if X2 is None:
            X2 = X
        X2 = self._slice(X2)
        r = self._unscaled_dist(X, X2) / self.lengthscale
        return self.variance * np.exp(-0.5 * r ** 2)

    def Kdiag(self, X):
        ret = np.empty(X.shape[0])
        ret[:] = self.variance
        return ret

    # --- GPy 1.9.5 Stationary/RBF gradient semantics (restated) ---
    def _scaled_dist(self, X, X2=None):
        if X2 is None:
            Xsq = np.sum(np.square(X), 1)
            r2 = -2.0 * np.dot(X, X.T) + (Xsq[:, None] + Xsq[None, :])
            r2[np.diag_indices(X.shape[0])] = 0.0
            r2 = np.clip(r2, 0, np.inf)
            return np.sqrt(r2) / self.lengthscale
        return self._unscaled_dist(X, X2) / self.lengthscale

    def _K_of_r(self, r):
        return self.variance * np.exp(-0.5 * r ** 2)

    @property
    def gradient(self):
        return np.hstack([np.ravel(self.variance.gradient), np.ravel(self.lengthscale.gradient)])

    @gradient.setter
    def gradient(self, g):
        g = np.ravel(g)
        self.variance.gradient = g[0]
        self.lengthscale.gradient = g[1]

    def update_gradients_full(self, dL_dK, X, X2=None):
        r = self._scaled_dist(X, X2)
        K = self._K_of_r(r)
        self.variance.gradient = np.sum(K * dL_dK) / self.variance
        dL_dr = (-r * K) * dL_dK
        self.lengthscale.gradient = -np.sum(dL_dr * r) / self.lengthscale

    def update_gradients_diag(self, dL_dKdiag, X):
        self.variance.gradient = np.sum(dL_dKdiag)
        self.lengthscale.gradient = 0.0

    def gradients_X(self, dL_dK, X, X2=None):
        r = self._scaled_dist(X, X2)
        invdist = 1.0 / np.where(r != 0.0, r, np.inf)
        dL_dr = (-r * self._K_of_r(r)) * dL_dK
        tmp = invdist * dL_dr
        if X2 is None:
            tmp = tmp + tmp.T
            X2 = X
        grad = np.empty(X.shape, dtype=np.float64)
        for q in range(self.input_dim):
            np.sum(tmp * (X[:, q][:, None] - X2[:, q][None, :]), axis=1, out=grad[:, q])
        return grad / self.lengthscale ** 2


class _Prod(object):
    def __iadd__(self, other):
        return self

    def __add__(self, other):
        return self


class _W(np.ndarray):
    pass


class Coregionalize(object):
    def __init__(self, input_dim, output_dim, rank=1, W=None, kappa=None, name="coregion"):
        self.input_dim, self.output_dim, self.rank = input_dim, output_dim, rank
        self.W = Param("W", np.array(W, dtype=float).reshape(output_dim, rank))
        self.kappa = Param("kappa", np.array(kappa, dtype=float).reshape(output_dim))
        self.name = name

    @property
    def gradient(self):
        return np.hstack([np.ravel(self.W.gradient), np.ravel(self.kappa.gradient)])

    @gradient.setter
    def gradient(self, g):
        g = np.ravel(g)
        n = self.W.size
        self.W.gradient = g[:n]
        self.kappa.gradient = g[n:]

    @property
    def B(self):
        W = np.asarray(self.W)
        return np.dot(W, W.T) + np.diag(np.asarray(self.kappa))


class Posterior(object):
    """GPy 1.9.5 `inference.latent_function_inference.posterior.Posterior` as the reference uses it (constructed from
    mean / cov / K at svmogp_inf.py:48-51,181; read lazily by the predict methods, svmogp.py:238-251,267-276): nothing is
    computed until a property is asked for.  `K_chol = jitchol(K)`; `woodbury_vector = dpotrs(K_chol, mean - prior_mean)`
    (= K^-1 mean); `woodbury_inv[:, :, i] = K^-1 (K - cov)[:, :, i] K^-1` by two dpotrs per slice of the
    `atleast_3d` difference -- so it is (N, N, 1) for a 2-D covariance.  Restated from GPy 1.9.5."""

    rungs = None     # make_golden hooks a list in here to record the jitter rung every K_chol took

    def __init__(self, mean=None, cov=None, K=None, prior_mean=0, **kw):
        self.mean, self.covariance, self._K, self.prior_mean = mean, cov, K, prior_mean
        self._K_chol = self._woodbury_vector = self._woodbury_inv = None

    @property
    def K_chol(self):
        if self._K_chol is None:
            self._K_chol = jitchol(self._K, _record=Posterior.rungs)
        return self._K_chol

    @property
    def woodbury_vector(self):
        if self._woodbury_vector is None:
            self._woodbury_vector, _ = dpotrs(self.K_chol, self.mean - self.prior_mean)
        return self._woodbury_vector

    @property
    def woodbury_inv(self):
        if self._woodbury_inv is None:
            B = np.atleast_3d(self._K) - np.atleast_3d(self.covariance)
            self._woodbury_inv = np.empty_like(B)
            for i in range(B.shape[-1]):
                tmp, _ = dpotrs(self.K_chol, B[:, :, i])
                self._woodbury_inv[:, :, i], _ = dpotrs(self.K_chol, tmp.T)
        return self._woodbury_inv


class LatentFunctionInference(object):
    pass


class Adadelta(object):
    """climin 0.1a1 `Adadelta` as util.py:327-329 drives it (restated): in place on `wrt`, `minimize_until(cb)`
    stops when `cb(info)` is true, `info['n_iter']` 1-based."""

    def __init__(self, wrt, fprime, step_rate=1, decay=0.9, momentum=0, offset=1e-4, args=None):
        self.wrt, self.fprime = wrt, fprime
        self.step_rate, self.decay, self.momentum, self.offset = step_rate, decay, momentum, offset
        self.gms, self.sms, self.step = np.zeros_like(wrt), np.zeros_like(wrt), np.zeros_like(wrt)
        self.n_iter = 0

    def __iter__(self):
        while True:
            d, o, m = self.decay, self.offset, self.momentum
            step1 = self.step * m
            self.wrt -= step1
            gradient = self.fprime(self.wrt)
            self.gms = (d * self.gms) + (1 - d) * gradient ** 2
            step2 = np.sqrt(self.sms + o) / np.sqrt(self.gms + o) * gradient * self.step_rate
            self.wrt -= step2
            self.step = step1 + step2
            self.sms = (d * self.sms) + (1 - d) * self.step ** 2
            self.n_iter += 1
            yield dict(n_iter=self.n_iter, gradient=gradient, args=(), kwargs={})

    def minimize_until(self, criterions):
        criterions = criterions if isinstance(criterions, (list, tuple)) else [criterions]
        for info in self:
            if any(c(info) for c in criterions):
                return info


def install():
    """Inject the stand-in modules. Idempotent."""
    if "GPy" in sys.modules and getattr(sys.modules["GPy"], "_hetmogp_standin", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    linalg = mod("GPy.util.linalg", jitchol=jitchol, dpotri=dpotri, dpotrs=dpotrs)
    chol = mod("GPy.util.choleskies", flat_to_triang=flat_to_triang, triang_to_flat=triang_to_flat)
    misc = mod("GPy.util.misc", safe_exp=safe_exp, safe_square=safe_square, kmm_init=None)
    from scipy.stats import norm as _norm
    ug = mod("GPy.util.univariate_Gaussian", std_norm_pdf=_norm.pdf, std_norm_cdf=_norm.cdf)
    util = mod("GPy.util", linalg=linalg, choleskies=chol, misc=misc, univariate_Gaussian=ug)
    link = mod("GPy.likelihoods.link_functions", Identity=Identity)
    liks = mod("GPy.likelihoods", link_functions=link, Likelihood=Likelihood)
    kern = mod("GPy.kern", RBF=RBF, Coregionalize=Coregionalize)
    post = mod("GPy.inference.latent_function_inference.posterior", Posterior=Posterior)
    lfi = mod("GPy.inference.latent_function_inference", LatentFunctionInference=LatentFunctionInference,
              posterior=post)
    inf = mod("GPy.inference", latent_function_inference=lfi)
    prm = mod("GPy.core.parameterization.param", Param=Param)
    prmz = mod("GPy.core.parameterization", param=prm, Param=Param)
    core = mod("GPy.core", SparseGP=SparseGP, parameterization=prmz, Param=Param)
    pl_util = mod("GPy.plotting.matplot_dep.util", fixed_inputs=None)
    pl_mpl = mod("GPy.plotting.matplot_dep", util=pl_util)
    pl = mod("GPy.plotting", matplot_dep=pl_mpl)
    gpy = mod("GPy", util=util, likelihoods=liks, kern=kern, inference=inf, core=core, plotting=pl,
              _hetmogp_standin=True)
    mod("climin", Adadelta=Adadelta)
    # scipy.misc.logsumexp / np.int were removed from the installed SciPy / NumPy
    import scipy.special
    if "scipy.misc" not in sys.modules:
        try:
            import scipy.misc  # noqa: F401
        except Exception:
            mod("scipy.misc")
    sm = sys.modules["scipy.misc"]
    if not hasattr(sm, "logsumexp"):
        sm.logsumexp = scipy.special.logsumexp
    if not hasattr(np, "int"):
        np.int = int
    return gpy
