#!/usr/bin/env python
"""The reference's notebooks/demo.ipynb flow (cells 1-8) on hetmogp_amd -- only the imports differ.

T = 2 [Gaussian(sigma=1), Bernoulli], N = 600 / 500 with a 99-row gap held out of task 2, M = 8 inducing points,
Q = 2 latent GPs, lengthscale 0.05, variance 0.5, 5 VEM iterations (L-BFGS-B E / M steps).  Run on an MI355X:
    python examples/demo.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hetmogp_amd import SVMOGP, HetLikelihood, Gaussian, Bernoulli, util  # noqa: E402
from hetmogp_amd.util import vem_algorithm as VEM  # noqa: E402


def main(seed=0, vem_iters=5, verbose=True):
    np.random.seed(seed)
    M, Q = 8, 2
    likelihood = HetLikelihood([Gaussian(sigma=1.), Bernoulli()])
    Y_metadata = likelihood.generate_metadata()
    X1 = np.sort(np.random.rand(600))[:, None]
    X2 = np.sort(np.random.rand(500))[:, None]
    X = [X1, X2]

    def true_u(x):                                      # demo.ipynb cell 2
        return np.hstack([4.5 * np.cos(2 * np.pi * x + 1.5 * np.pi) - 3 * np.sin(4.3 * np.pi * x + 0.3 * np.pi)
                          + 5 * np.cos(7 * np.pi * x + 2.4 * np.pi),
                          4.5 * np.cos(1.5 * np.pi * x + 0.5 * np.pi) + 5 * np.sin(3 * np.pi * x + 1.5 * np.pi)
                          - 5.5 * np.cos(8 * np.pi * x + 0.25 * np.pi)])
    Wtrue = [np.array([[-0.5], [0.1]]), np.array([[-0.1], [0.6]])]
    trueF = [sum(Wtrue[q][d] * true_u(X[d])[:, q, None] for q in range(2)) for d in range(2)]
    Y = likelihood.samples(F=trueF, Y_metadata=Y_metadata)
    gap = np.r_[351:450]                                # cell 5: held-out gap of the binary task
    X2test, Y2test = X[1][gap], Y[1][gap]
    X = [X1, np.delete(X2, gap, 0)]
    Y = [Y[0], np.delete(Y[1], gap, 0)]

    kern_list = util.latent_functions_prior(Q, lenghtscale=np.array([.05] * Q), variance=np.array([.5] * Q), input_dim=1)
    Z = np.linspace(0, 1, M)[:, None]
    model = SVMOGP(X=X, Y=Y, Z=Z, kern_list=kern_list, likelihood=likelihood, Y_metadata=Y_metadata)
    e0 = float(model.log_likelihood()[0, 0])
    model = VEM(model, stochastic=False, vem_iters=vem_iters, optZ=True, verbose=False, verbose_plot=False, non_chained=True)
    e1 = float(model.log_likelihood()[0, 0])
    m_gap, v_gap = model.predictive_new(np.sort(X2test, 0), output_function_ind=1)
    p_gap = 1.0 / (1.0 + np.exp(-m_gap))
    acc = float(np.mean((p_gap > 0.5) == (Y2test > 0.5)))
    if verbose:
        print("ELBO before VEM %.2f, after %d VEM iterations %.2f; gap accuracy of the Bernoulli task %.2f" % (e0, vem_iters, e1, acc))
    return e0, e1, acc


if __name__ == "__main__":
    main()
