"""hetmogp_amd -- MI355X-native (gfx950) engine for the svmogp_inf ELBO path of pmorenoz/HetMOGP.

The compute path is `libhetmogp_hip.so` (hand-written HIP, C ABI in include/hetmogp_hip.h).  Importing this
package without the built library raises: there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises if libhetmogp_hip.so is missing)
from .engine import Engine  # noqa: F401
from .svmogp import SVMOGP, HetMOGP  # noqa: F401
from .likelihoods import (HetLikelihood, Gaussian, Bernoulli, HetGaussian, Categorical, Poisson, Exponential, Gamma,  # noqa: F401
                          Beta)
from .util import vem_algorithm, latent_functions_prior, random_W_kappas, LCM  # noqa: F401
