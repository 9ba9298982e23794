"""Minimal parameter containers with the slice of the paramz surface the reference touches
(hetmogp/svmogp.py:66-75,106-113,153-166; hetmogp/util.py:285-317): ndarray values, a `.gradient` of the same shape
that basic slices write through to, `.fix()` / `.unfix()` / `.is_fixed`, and the Logexp positive transform paramz
applies to `variance`, `lengthscale` and `kappa` in the optimiser's view.  paramz itself is not a dependency."""
import re
import weakref

import numpy as np


class Param(np.ndarray):
    def __new__(cls, name, input_array, positive=False, storage=None):
        src = np.atleast_1d(np.asarray(input_array, dtype=np.float64))
        if storage is not None:      # caller-provided buffer of the same shape (e.g. page-locked: engine.pinned_empty)
            storage[...] = src
            obj = storage.view(cls)
        else:
            obj = np.array(src, dtype=np.float64).view(cls)                   # copy: detached from the caller
        obj.name = name
        obj._grad = np.zeros(obj.shape)
        obj._fixed = [False]
        obj._observers = []              # shared with every view: callables fired after a write (paramz's notification)
        obj.positive = positive
        return obj

    def __array_finalize__(self, obj):
        self.name = getattr(obj, "name", None)
        self._grad = None
        self._fixed = getattr(obj, "_fixed", [False])
        self._observers = getattr(obj, "_observers", [])
        self.positive = getattr(obj, "positive", False)

    # paramz re-runs the model's parameters_changed() on every write to a parameter; here a write notifies the
    # observers (the model marks itself dirty and re-evaluates lazily before the next read of a derived quantity).
    # Writes that notify: item / slice assignment (`p[...] = v`, `p[i, j] = v`) and the in-place operators
    # (`+= -= *= /= **= //= %=`).  Writes that go around this class -- `p.values[...] = v`, `np.copyto(p, v)`,
    # `ufunc(..., out=p)`, a plain-ndarray view from `np.asarray(p)` -- do NOT: follow them with `model.touch()`.
    # Observers that are bound methods are held weakly: a kernel object reused in a second model neither keeps the
    # first model alive nor notifies a dead one.
    def add_observer(self, fn):
        try:
            ref = weakref.WeakMethod(fn)
        except TypeError:            # a plain function / lambda: held strongly
            ref = (lambda f: (lambda: f))(fn)
        self._observers.append(ref)

    def _notify(self):
        dead = []
        for ref in self._observers:
            fn = ref()
            if fn is None:
                dead.append(ref)
            else:
                fn(self)
        for ref in dead:
            self._observers.remove(ref)

    def __setitem__(self, idx, val):
        np.ndarray.__setitem__(self, idx, val)
        self._notify()

    def _inplace(self, op, other):
        out = op(np.asarray(self), other)      # acts on the shared buffer
        self._notify()
        return self

    def __iadd__(self, o):
        return self._inplace(np.ndarray.__iadd__, o)

    def __isub__(self, o):
        return self._inplace(np.ndarray.__isub__, o)

    def __imul__(self, o):
        return self._inplace(np.ndarray.__imul__, o)

    def __itruediv__(self, o):
        return self._inplace(np.ndarray.__itruediv__, o)

    def __ipow__(self, o):
        return self._inplace(np.ndarray.__ipow__, o)

    def __ifloordiv__(self, o):
        return self._inplace(np.ndarray.__ifloordiv__, o)

    def __imod__(self, o):
        return self._inplace(np.ndarray.__imod__, o)

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
        self._grad[...] = np.asarray(val, dtype=np.float64).reshape(self._grad.shape)

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


class ParamGroup(object):
    """What `model['.*.lengthscale']` returns: fix()/unfix() over every matching parameter (util.py:285-317)."""

    def __init__(self, params):
        self.params = list(params)

    def fix(self):
        for p in self.params:
            p.fix()

    def unfix(self):
        for p in self.params:
            p.unfix()

    def __len__(self):
        return len(self.params)


def match(named_params, pattern):
    rx = re.compile(pattern)
    return ParamGroup(p for n, p in named_params if rx.search(n) or rx.fullmatch(n))


# paramz Logexp: theta = log(1 + exp(x)); gradient factor d theta / d x = 1 - exp(-theta)
def logexp_f(x):
    # paramz Logexp.f clips its argument to [-36, 36]: the transformed value never reaches exactly 0
    return np.where(x > 36.0, x, np.log1p(np.exp(np.clip(x, -36.0, 36.0))))


def logexp_finv(theta):
    return np.where(theta > 36.0, theta, np.log(np.expm1(np.minimum(theta, 36.0))))


def logexp_gradfactor(theta):
    return np.where(theta > 36.0, 1.0, -np.expm1(-theta))
