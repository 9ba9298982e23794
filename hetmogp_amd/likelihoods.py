"""Likelihood descriptors with the reference's class names and constructor arguments (likelihoods/*.py) and the
heterogeneous wrapper (hetmogp/het_likelihood.py).  They describe the model to the engine; the variational
expectations themselves are HIP kernels (csrc/lik_device.h).  `var_exp` / `var_exp_derivatives` are offered with the
reference's signatures and run on the device."""
import numpy as np


class _Lik(object):
    name = None
    _dims = (1, 1, 1)

    def kwargs(self):
        return {}

    def get_metadata(self):
        """(dim_y, dim_f, dim_p) as the reference's get_metadata()."""
        return self._dims

    def ismulti(self):
        return False

    def var_exp(self, Y, m, v, gh_points=None, Y_metadata=None):
        from .engine import var_exp
        ve, _, _ = var_exp(self.name, Y, m, v, **self.kwargs())
        return ve[:, None]

    def var_exp_derivatives(self, Y, m, v, gh_points=None, Y_metadata=None):
        from .engine import var_exp
        _, dm, dv = var_exp(self.name, Y, m, v, **self.kwargs())
        return dm, dv

    def log_predictive(self, Ytest, mu_F_star, v_F_star, num_samples, seed=0):
        """The reference's Monte-Carlo log predictive, including its 1/num_samples factor on the sum over test points
        (e.g. bernoulli.py:130-144).  Sampling happens on the device."""
        from .engine import log_predictive_rows
        lp = log_predictive_rows(self.name, Ytest, mu_F_star, v_F_star, num_samples, seed, **self.kwargs())
        return (1.0 / num_samples) * lp.sum()

    def predictive(self, m, v, gh_points=None, Y_metadata=None):
        """Predictive mean / variance of y (the reference's `predictive`; Gauss-Hermite order of a fresh instance)."""
        from .engine import predictive
        return predictive(self.name, m, v, **self.kwargs())

    def samples(self, F, num_samples=1, Y_metadata=None, seed=None):
        """One draw y ~ p(y | F[n]) per row, (N, 1) -- the reference's `samples` (e.g. gaussian.py:36-39, gamma.py:43-50,
        categorical.py:65-75: same link functions and clips, labels 1..K), generated on the device (`hmogp_sample`).
        The reference draws from NumPy's global generator; here the device generator is keyed by `seed`, which is itself
        drawn from NumPy's global generator when not given -- so `np.random.seed(k)` still makes a run reproducible, but
        the stream differs from the reference's (only the distribution is the same)."""
        from .engine import sample
        if seed is None:       # NB: one extra draw from NumPy's GLOBAL generator per call (shifts it relative to the reference)
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        F = np.asarray(F, dtype=float)
        y = sample(self.name, F, seed=seed, **self.kwargs())
        if self.get_metadata()[1] == 1 and F.ndim == 2 and F.shape[1] > 1:
            return y.reshape(F.shape)      # one-function likelihoods: every entry of F is a draw, shape kept (gaussian.py:36-39)
        return y


class Gaussian(_Lik):
    name = "Gaussian"

    def __init__(self, sigma=None, gp_link=None):
        self.sigma = 0.5 if sigma is None else sigma       # gaussian.py:21-24

    def kwargs(self):
        return {"sigma": self.sigma}


class Bernoulli(_Lik):
    name = "Bernoulli"

    def __init__(self, gp_link=None):
        pass


class HetGaussian(_Lik):
    name = "HetGaussian"
    _dims = (1, 2, 1)

    def __init__(self, gp_link=None):
        pass


class Poisson(_Lik):
    name = "Poisson"

    def __init__(self, gp_link=None):
        pass


class Exponential(_Lik):
    name = "Exponential"

    def __init__(self, gp_link=None):
        pass


class Gamma(_Lik):
    name = "Gamma"
    _dims = (1, 2, 1)

    def __init__(self, gp_link=None):
        pass


class Beta(_Lik):
    name = "Beta"
    _dims = (1, 2, 1)

    def __init__(self, gp_link=None):
        pass


class Categorical(_Lik):
    name = "Categorical"

    def __init__(self, K, gp_link=None):
        self.K = int(K)

    def kwargs(self):
        return {"K": self.K}

    def get_metadata(self):
        return 1, self.K - 1, self.K - 1                   # categorical.py:287-291


class HetLikelihood(object):
    """het_likelihood.py:10-44,85-90."""

    def __init__(self, likelihoods_list, gp_link=None, name="heterogeneous_likelihood"):
        self.likelihoods_list = list(likelihoods_list)
        self.name = name

    def generate_metadata(self):
        t_index = np.arange(len(self.likelihoods_list))
        y_index, f_index, d_index, p_index = [], [], [], []
        for t, lik in enumerate(self.likelihoods_list):
            dim_y, dim_f, dim_p = lik.get_metadata()
            y_index += [t] * dim_y
            f_index += [t] * dim_f
            d_index += list(range(dim_f))
            p_index += [t] * dim_p
        return {"task_index": t_index, "y_index": np.int_(y_index), "function_index": np.int_(f_index),
                "d_index": np.int_(d_index), "pred_index": np.int_(p_index)}

    def num_output_functions(self, Y_metadata):
        return Y_metadata["function_index"].flatten().shape[0]

    def ismulti(self, task):
        return self.likelihoods_list[task].ismulti()

    def specs(self):
        return [(l.name, l.kwargs()) for l in self.likelihoods_list]

    def samples(self, F, Y_metadata):
        """het_likelihood.py:72-83: one draw per task from its likelihood, generated on the device."""
        return [l.samples(F[t], num_samples=1) for t, l in enumerate(self.likelihoods_list)]

    def var_exp(self, Y, mu_F, v_F, Y_metadata):
        return [l.var_exp(Y[t], mu_F[t], v_F[t]) for t, l in enumerate(self.likelihoods_list)]

    def predictive(self, mu_F_pred, v_F_pred, Y_metadata):
        """het_likelihood.py:133-148."""
        out = [l.predictive(mu_F_pred[t], v_F_pred[t]) for t, l in enumerate(self.likelihoods_list)]
        return [o[0] for o in out], [o[1] for o in out]

    def negative_log_predictive(self, Ytest, mu_F_star, v_F_star, Y_metadata, num_samples, seed=0):
        """het_likelihood.py:150-164."""
        logpred = 0.0
        for t, l in enumerate(self.likelihoods_list):
            logpred += l.log_predictive(Ytest[t], mu_F_star[t], v_F_star[t], num_samples, seed=seed + t)
        return -logpred

    def var_exp_derivatives(self, Y, mu_F, v_F, Y_metadata):
        out = [l.var_exp_derivatives(Y[t], mu_F[t], v_F[t]) for t, l in enumerate(self.likelihoods_list)]
        return [o[0] for o in out], [o[1] for o in out]
