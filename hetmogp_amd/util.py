"""Host-side helpers with the reference's names and calling conventions (hetmogp/util.py): model construction
(`latent_functions_prior`, `random_W_kappas`, `LCM`), the contiguous minibatch slicer (`mini_slices`,
`draw_mini_slices`) and the VEM driver (`vem_algorithm`: L-BFGS-B alternation or Adadelta SVI).  No arithmetic of the
ELBO lives here -- that is `SVMOGP.parameters_changed()` -> libhetmogp_hip.so.  The data generators of the reference's
notebook (util.py:21-50) are not part of the path; `hetmogp_amd/synthetic.py` builds the benchmark inputs."""
import itertools
import random
from functools import partial

import numpy as np

from .kern import RBF, Coregionalize


def mini_slices(n_samples, batch_size):
    """Contiguous minibatch slices covering `n_samples` rows; the last one is short when the size does not divide
    (behaviour of util.py:52-59 -- the slice stops are NOT clipped, exactly as there)."""
    return [slice(start, start + batch_size) for start in range(0, n_samples, batch_size)]


def draw_mini_slices(n_samples, batch_size, with_replacement=False):
    """Endless generator over `mini_slices` (util.py:62-72).  Without replacement the reference visits the slices in
    their natural order on every epoch (its per-epoch shuffle acts on a temporary); with replacement it yields ONE
    random slice and stops.  Both behaviours, and the draws they take from the `random` module, are kept."""
    slices = mini_slices(n_samples, batch_size)
    if with_replacement:
        yield random.choice(slices)
        return
    for epoch in itertools.count():
        scratch = list(range(len(slices)))
        random.shuffle(scratch)          # consumes the module RNG like the reference does; the order is not used
        yield from slices


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
    """Coregionalisation weights of util.py:92-103: per latent q a (D, 1) column whose entries are N(+0.5, 0.5^2) or
    N(-0.5, 0.5^2) with equal probability, divided by sqrt(rank); kappa = 0 (always fixed).  Draw order per q:
    the D coin flips, the D positive-mean normals, the D negative-mean normals (same NumPy stream as the reference)."""
    W_list, kappa_list = [], []
    for _ in range(Q):
        heads = np.random.binomial(n=1, p=np.full((D, 1), 0.5)).astype(bool)
        around_plus = np.random.normal(loc=0.5, scale=0.5, size=(D, 1))
        around_minus = np.random.normal(loc=-0.5, scale=0.5, size=(D, 1))
        W_list.append(np.where(heads, around_plus, around_minus) / np.sqrt(rank))
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


# The alternation of util.py:284-331 as data: which parameter groups are frozen in the variational E-step and which in
# the M-step.  `Z` follows optZ and `W` follows non_chained in the M-step (they stay frozen otherwise).
_GROUPS = {
    "lengthscale": lambda m: m[".*.lengthscale"],
    "variance": lambda m: m[".*.variance"],
    "W": lambda m: m[".*.W"],
    "kappa": lambda m: m[".*.kappa"],
    "Z": lambda m: m.Z,
    "m_u": lambda m: m.q_u_means,
    "L_u": lambda m: m.q_u_chols,
}
VEM_SCHEDULE = (
    # (label, groups frozen in this half-step, groups released in this half-step)
    ("VE", ("lengthscale", "variance", "Z", "W"), ("m_u", "L_u")),
    ("VM", ("m_u", "L_u"), ("lengthscale", "variance", "Z", "W")),
)


def _apply_half_step(model, frozen, released, optZ, non_chained):
    for g in frozen:
        _GROUPS[g](model).fix()
    for g in released:
        if (g == "Z" and not optZ) or (g == "W" and not non_chained):
            continue
        _GROUPS[g](model).unfix()


def vem_algorithm(model, stochastic=False, vem_iters=None, step_rate=None, verbose=False, optZ=True, verbose_plot=False,
                  non_chained=True, device_optimizer=True, qu_optimizer="adadelta", natgrad_gamma=0.1):
    """Variational EM driver with the signature of util.py:284-331.  Batch mode alternates L-BFGS-B over q(u) (VE) and
    over the hyper-parameters (VM) following VEM_SCHEDULE, at most 100 iterations each, and reports the ELBO after every
    half-step; stochastic mode runs Adadelta (momentum 0.9) on `model.stochastic_grad` for `vem_iters` iterations.
    lengthscale and kappa start frozen; kappa is never released.  `device_optimizer` (stochastic mode, no reference
    equivalent): keep q(u) and its Adadelta state in HBM (`SVMOGP.device_adadelta`); the iterates are bit-identical to
    the host optimiser's.  `qu_optimizer="natgrad"` (stochastic mode, no reference equivalent; the north-star names it): the
    E-steps move q(u) by NATURAL-gradient steps of size `natgrad_gamma` on the device-resident q(u) (`hmogp_qu_natgrad`)
    instead of Adadelta on its Euclidean gradient; the hyper-parameters keep Adadelta on the M-steps."""
    if qu_optimizer not in ("adadelta", "natgrad"):
        raise ValueError("qu_optimizer must be 'adadelta' or 'natgrad'")
    vem_iters = 5 if vem_iters is None else vem_iters
    _GROUPS["lengthscale"](model).fix()
    _GROUPS["kappa"](model).fix()
    if not stochastic:
        model.elbo = np.empty((vem_iters, 1))
        for it in range(1, vem_iters + 1):
            for label, frozen, released in VEM_SCHEDULE:
                _apply_half_step(model, frozen, released, optZ, non_chained)
                model.optimize(messages=verbose, max_iters=100)
                print("VEM %d/%d  %s-step  ELBO = %s" % (it, vem_iters, label, model.log_likelihood().flatten()))
        return model
    rate = 0.01 if step_rate is None else step_rate
    model.elbo = np.empty((vem_iters + 1, 1))
    stop = partial(model.callback, max_iter=vem_iters, verbose=verbose, verbose_plot=verbose_plot)
    optimizer = None
    if qu_optimizer == "natgrad":
        make = getattr(model, "device_natgrad", None)
        optimizer = make(gamma=natgrad_gamma, step_rate=rate, momentum=0.9) if make is not None else None
        if optimizer is None:
            raise ValueError("qu_optimizer='natgrad' needs a stochastic model with a free q(u)")
    elif device_optimizer and getattr(model, "device_adadelta", None) is not None:
        optimizer = model.device_adadelta(step_rate=rate, momentum=0.9)      # None when it does not apply
    if optimizer is None:
        optimizer = Adadelta(model.optimizer_array, model.stochastic_grad, step_rate=rate, momentum=0.9)
    optimizer.minimize_until(stop)
    return model
