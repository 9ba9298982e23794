"""Parameter holders with the names the reference's model wiring uses: `RBF` (GPy.kern.RBF as built by
util.latent_functions_prior, util.py:75-90) and `Coregionalize` (GPy.kern.Coregionalize with rank 1, util.py:120).
They hold values and gradients only; K_uu / K_uf are built on the GPU (csrc/rowpass.hip)."""
import numpy as np

from .param import Param


class RBF(object):
    def __init__(self, input_dim, variance=1.0, lengthscale=None, ARD=False, name="rbf"):
        if ARD:
            raise NotImplementedError("the reference builds isotropic RBF kernels only (util.py:87)")
        self.input_dim = int(input_dim)
        self.variance = Param("variance", np.atleast_1d(variance).astype(float)[:1], positive=True)
        self.lengthscale = Param("lengthscale", np.atleast_1d(1.0 if lengthscale is None else lengthscale).astype(float)[:1],
                                 positive=True)
        self.name = name

    def copy(self):
        return RBF(self.input_dim, self.variance.values.copy(), self.lengthscale.values.copy(), name=self.name)

    @property
    def gradient(self):
        return np.hstack([np.ravel(self.variance.gradient), np.ravel(self.lengthscale.gradient)])

    @gradient.setter
    def gradient(self, g):
        g = np.ravel(g)
        self.variance.gradient = g[0]
        self.lengthscale.gradient = g[1]

    def K(self, X, X2=None):
        """GPy RBF.K on the device (host round trip; for plotting / small predictions only)."""
        from .engine import rbf_cross_cov
        X2 = X if X2 is None else X2
        return rbf_cross_cov(X, X2, float(self.variance[0]), float(self.lengthscale[0]))

    def Kdiag(self, X):
        return np.full(np.asarray(X).shape[0], float(self.variance[0]))


class Coregionalize(object):
    def __init__(self, input_dim, output_dim, rank=1, W=None, kappa=None, name="coregion"):
        if rank != 1:
            raise NotImplementedError("the reference fixes rank = 1 (svmogp.py:28,62)")
        self.input_dim, self.output_dim, self.rank = input_dim, output_dim, rank
        W = 0.5 * np.random.randn(output_dim, rank) / np.sqrt(rank) if W is None else W
        kappa = 0.5 * np.ones(output_dim) if kappa is None else kappa
        self.W = Param("W", np.asarray(W, dtype=float).reshape(output_dim, rank))
        self.kappa = Param("kappa", np.asarray(kappa, dtype=float).reshape(output_dim), positive=True)
        self.name = name

    @property
    def B(self):
        W = self.W.values
        return W @ W.T + np.diag(self.kappa.values)

    @property
    def gradient(self):
        return np.hstack([np.ravel(self.W.gradient), np.ravel(self.kappa.gradient)])

    @gradient.setter
    def gradient(self, g):
        g = np.ravel(g)
        n = self.W.size
        self.W.gradient = g[:n]
        self.kappa.gradient = g[n:]
