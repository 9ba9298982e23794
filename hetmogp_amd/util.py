"""Host-side helpers with the reference's names and calling conventions (hetmogp/util.py): model construction
(`latent_functions_prior`, `random_W_kappas`, `LCM`), the contiguous minibatch slicer (`mini_slices`,
`draw_mini_slices`) and the VEM driver (`vem_algorithm`: L-BFGS-B alternation or Adadelta SVI).  No arithmetic of the
ELBO lives here -- that is `SVMOGP.parameters_changed()` -> libhetmogp_hip.so."""
import random
from functools import partial

import numpy as np

from .kern import RBF, Coregionalize


def get_batch_scales(X_all, X):
    """util.py:15-19."""
    return [float(xa.shape[0]) / float(x.shape[0]) for xa, x in zip(X_all, X)]


def true_u_functions(X_list, Q):
    """util.py:21-35: random three-sinusoid latent functions evaluated at each task's inputs."""
    amplitude = (1.5 - 0.5) * np.random.rand(Q, 3) + 0.5
    freq = (3 - 1) * np.random.rand(Q, 3) + 1
    shift = 2 * np.random.rand(Q, 3)
    u_functions = []
    for X in X_list:
        u_task = np.empty((X.shape[0], Q))
        for q in range(Q):
            u_task[:, q, None] = 3 * amplitude[q, 0] * np.cos(freq[q, 0] * np.pi * X + shift[q, 0] * np.pi) - \
                2 * amplitude[q, 1] * np.sin(2 * freq[q, 1] * np.pi * X + shift[q, 1] * np.pi) + \
                amplitude[q, 2] * np.cos(4 * freq[q, 2] * np.pi * X + shift[q, 2] * np.pi)
        u_functions.append(u_task)
    return u_functions


def true_f_functions(true_u, W_list, D, likelihood_list, Y_metadata):
    """util.py:37-50: F_t[:, j] = sum_q W_q[d] u_q for the functions d of task t."""
    f_index = Y_metadata["function_index"].flatten()
    d_index = Y_metadata["d_index"].flatten()
    true_f = []
    for t, u_task in enumerate(true_u):
        _, num_f_task, _ = likelihood_list[t].get_metadata()
        F = np.zeros((u_task.shape[0], num_f_task))
        for q, W in enumerate(W_list):
            for d in range(D):
                if f_index[d] == t:
                    F[:, d_index[d], None] += np.tile(W[d].T, (u_task.shape[0], 1)) * u_task[:, q, None]
        true_f.append(F)
    return true_f


def mini_slices(n_samples, batch_size):
    """util.py:52-59: contiguous slices, the last one may be short."""
    n_batches, rest = divmod(n_samples, batch_size)
    if rest != 0:
        n_batches += 1
    return [slice(i * batch_size, (i + 1) * batch_size) for i in range(n_batches)]


def draw_mini_slices(n_samples, batch_size, with_replacement=False):
    """util.py:62-72.  The reference shuffles a temporary copy of the index list (`random.shuffle(list(idxs))`), so the
    slices are always visited in order; reproduced as is."""
    slices = mini_slices(n_samples, batch_size)
    idxs = list(range(len(slices)))
    if with_replacement:
        yield random.choice(slices)
    else:
        while True:
            random.shuffle(list(idxs))
            for i in idxs:
                yield slices[i]


def latent_functions_prior(Q, lenghtscale=None, variance=None, input_dim=None):
    """util.py:75-90 (the misspelt keyword is the reference's)."""
    lenghtscale = np.random.rand(Q) if lenghtscale is None else lenghtscale
    variance = np.random.rand(Q) if variance is None else variance
    kern_list = []
    for q in range(Q):
        k = RBF(input_dim=input_dim, lengthscale=lenghtscale[q], variance=variance[q], name="rbf")
        k.name = "kern_q" + str(q)
        kern_list.append(k)
    return kern_list


def random_W_kappas(Q, D, rank, experiment=False):
    """util.py:92-103: W = +/- N(0.5, 0.5^2) / sqrt(rank), kappa = 0."""
    W_list, kappa_list = [], []
    for q in range(Q):
        p = np.random.binomial(n=1, p=0.5 * np.ones((D, 1)))
        Ws = p * np.random.normal(loc=0.5, scale=0.5, size=(D, 1)) - (p - 1) * np.random.normal(loc=-0.5, scale=0.5, size=(D, 1))
        W_list.append(Ws / np.sqrt(rank))
        kappa_list.append(np.zeros(D))
    return W_list, kappa_list


def LCM(input_dim, output_dim, kernels_list, W_list, kappa_list, rank, name="B_q"):
    """util.py:126-143: one Coregionalize object per latent GP (the product kernel K itself is never evaluated by the
    reference's inference; None is returned in its place)."""
    B_q = []
    for q in range(len(kernels_list)):
        B = Coregionalize(input_dim=input_dim, output_dim=output_dim, rank=rank, W=W_list[q], kappa=kappa_list[q])
        B.name = "%s%s" % (name, q)
        B_q.append(B)
    return None, B_q


class Adadelta(object):
    """climin.Adadelta as the reference calls it (util.py:327): wrt is updated IN PLACE.
    step1 = m*step; wrt -= step1; g = f'(wrt); gms = d*gms + (1-d) g^2; step2 = sqrt(sms+o)/sqrt(gms+o) * g * rate;
    wrt -= step2; step = step1 + step2; sms = d*sms + (1-d) step^2."""

    def __init__(self, wrt, fprime, step_rate=1.0, decay=0.9, momentum=0.0, offset=1e-4):
        self.wrt, self.fprime = wrt, fprime
        self.step_rate, self.decay, self.momentum, self.offset = step_rate, decay, momentum, offset
        self.gms = np.zeros_like(wrt)
        self.sms = np.zeros_like(wrt)
        self.step = np.zeros_like(wrt)
        self.n_iter = 0

    _BLOCK = 32768      # elements per cache block (8 arrays x 256 KB)

    def __iter__(self):
        # Same operations in the same order as the formulas above (bit-identical iterates), but in place and cache-blocked:
        # at the sizes of the path (1.6 M parameters at M = 1024, Q = 3) twenty full-length NumPy passes cost more than a
        # gradient evaluation on the GPU; block by block the eight arrays stay in the host's L2.
        n = self.wrt.size
        wrt, gms, sms, step = (a.reshape(-1) for a in (self.wrt, self.gms, self.sms, self.step))
        step1 = np.empty(n)
        blocks = [(b, min(n, b + self._BLOCK)) for b in range(0, n, self._BLOCK)]
        s1, s2 = np.empty(self._BLOCK), np.empty(self._BLOCK)
        while True:
            d, o, m, rate = self.decay, self.offset, self.momentum, self.step_rate
            for b, e in blocks:
                np.multiply(step[b:e], m, out=step1[b:e])
                wrt[b:e] -= step1[b:e]
            g = self.fprime(self.wrt)
            gf = np.ascontiguousarray(g, dtype=np.float64).reshape(-1)
            for b, e in blocks:
                t1, t2, gb = s1[:e - b], s2[:e - b], gf[b:e]
                np.multiply(gb, gb, out=t1)                # gms = d * gms + (1 - d) * g ** 2
                t1 *= (1 - d)
                gms[b:e] *= d
                gms[b:e] += t1
                np.add(sms[b:e], o, out=t1)                # step2 = sqrt(sms + o) / sqrt(gms + o) * g * step_rate
                np.sqrt(t1, out=t1)
                np.add(gms[b:e], o, out=t2)
                np.sqrt(t2, out=t2)
                t1 /= t2
                t1 *= gb
                t1 *= rate
                wrt[b:e] -= t1
                np.add(step1[b:e], t1, out=step[b:e])      # step = step1 + step2
                np.multiply(step[b:e], step[b:e], out=t2)  # sms = d * sms + (1 - d) * step ** 2
                t2 *= (1 - d)
                sms[b:e] *= d
                sms[b:e] += t2
            self.n_iter += 1
            yield dict(n_iter=self.n_iter, gradient=g, step=self.step)

    def minimize_until(self, criterion):
        for info in self:
            if criterion(info):
                return info


def vem_algorithm(model, stochastic=False, vem_iters=None, step_rate=None, verbose=False, optZ=True, verbose_plot=False,
                  non_chained=True):
    """util.py:284-331."""
    model[".*.lengthscale"].fix()
    if vem_iters is None:
        vem_iters = 5
    model[".*.kappa"].fix()  # must be always fixed
    model.elbo = np.empty((vem_iters, 1))
    if stochastic is False:
        for i in range(vem_iters):
            # variational E-step
            model[".*.lengthscale"].fix()
            model[".*.variance"].fix()
            model.Z.fix()
            model[".*.W"].fix()
            model.q_u_means.unfix()
            model.q_u_chols.unfix()
            model.optimize(messages=verbose, max_iters=100)
            print("iteration (" + str(i + 1) + ") VE step, ELBO=" + str(model.log_likelihood().flatten()))
            # variational M-step
            model[".*.lengthscale"].unfix()
            model[".*.variance"].unfix()
            if optZ:
                model.Z.unfix()
            if non_chained:
                model[".*.W"].unfix()
            model.q_u_means.fix()
            model.q_u_chols.fix()
            model.optimize(messages=verbose, max_iters=100)
            print("iteration (" + str(i + 1) + ") VM step, ELBO=" + str(model.log_likelihood().flatten()))
    else:
        if step_rate is None:
            step_rate = 0.01
        sto_iters = vem_iters
        model.elbo = np.empty((sto_iters + 1, 1))
        optimizer = Adadelta(model.optimizer_array, model.stochastic_grad, step_rate=step_rate, momentum=0.9)
        c_full = partial(model.callback, max_iter=sto_iters, verbose=verbose, verbose_plot=verbose_plot)
        optimizer.minimize_until(c_full)
    return model
