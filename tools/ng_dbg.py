import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import hetmogp_amd as H
from hetmogp_amd.kern import RBF
from hetmogp_amd.synthetic import make_case
SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
N_all, B, M, Q, P = 1000000, 8192, 1024, 3, 1
prm, X, Y = make_case(SPECS, [N_all] * 4, M=M, Q=Q, P=P, seed=20260932)
lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson(), H.Gamma()])
np.random.seed(1)
kern2 = [RBF(P, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
model2 = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :P].copy(), kern_list=kern2, likelihood=lik,
                  Y_metadata=lik.generate_metadata(), batch_size=B)
model2[".*.lengthscale"].fix(); model2[".*.kappa"].fix(); model2.Z.fix(); model2.stochastic = True
gam = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
ng = model2.device_natgrad(gamma=gam, step_rate=0.005, momentum=0.9)
it = iter(ng)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    info = next(it)
    print(i, "elbo %.6g" % float(model2._log_marginal_likelihood[0, 0]), "gamma", info["gamma"], "rej", ng.rejected,
          "var", [float(k.variance[0]) for k in kern2], "vneg", model2._engine.last.get("v_negative"), flush=True)
