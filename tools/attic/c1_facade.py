"""Wall time of one parameters_changed() of the FACADE at BASELINE config 1 (what an optimiser loop of the reference pays per
objective evaluation), its cProfile, and the wall time of the notebook demo's 5 VEM iterations."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples"))
import numpy as np
from hetmogp_amd import SVMOGP, HetLikelihood, HetGaussian, Bernoulli, Categorical, util

np.random.seed(0)
M, Q, N = 50, 2, 1000
lik = HetLikelihood([HetGaussian(), Bernoulli(), Categorical(K=3)])
meta = lik.generate_metadata()
X = [np.sort(np.random.rand(N))[:, None] for _ in range(3)]
Y = [np.sin(6 * X[0]) + 0.1 * np.random.randn(N, 1), (np.sin(5 * X[1]) > 0).astype(float), np.floor(3 * X[2]).clip(0, 2)]
kern = util.latent_functions_prior(Q, lenghtscale=np.array([.05] * Q), variance=np.array([.5] * Q), input_dim=1)
model = SVMOGP(X=X, Y=Y, Z=np.linspace(0, 1, M)[:, None], kern_list=kern, likelihood=lik, Y_metadata=meta)
x = model.optimizer_array.copy()
for _ in range(300):
    model._grads(x)
t0 = time.perf_counter()
n = 1000
for _ in range(n):
    model._grads(x)
print("facade objective+gradient at C1: %.1f us per call" % (1e6 * (time.perf_counter() - t0) / n))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    model._grads(x)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
import demo
t0 = time.perf_counter()
demo.main(seed=0, vem_iters=5, verbose=False)
print("notebook demo, 5 VEM iterations: %.2f s" % (time.perf_counter() - t0))
