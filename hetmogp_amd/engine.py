"""NumPy-level wrapper of the C ABI (one object = one ``hmogp_handle``): the inner layer of the facade.

Mirrors what ``SVMOGP.parameters_changed`` needs from ``SVMOGPInf.inference`` + the gradient assembly
(reference: hetmogp/svmogp.py:85-166, hetmogp/svmogp_inf.py:23-250): parameters in, ELBO and parameter
gradients out.  All arithmetic happens in ``libhetmogp_hip.so``; this file only marshals arrays.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check

LIK_IDS = dict(Gaussian=_lib.LIK_GAUSSIAN, Bernoulli=_lib.LIK_BERNOULLI, HetGaussian=_lib.LIK_HETGAUSSIAN,
               Categorical=_lib.LIK_CATEGORICAL, Poisson=_lib.LIK_POISSON, Exponential=_lib.LIK_EXPONENTIAL,
               Gamma=_lib.LIK_GAMMA, Beta=_lib.LIK_BETA)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_lib.c_double_p)


def lik_dim_f(name, **kw):
    """Number of latent parameter functions (the reference's ``get_metadata()[1]``)."""
    if name == "Categorical":
        return int(kw["K"]) - 1
    return dict(Gaussian=1, Bernoulli=1, HetGaussian=2, Poisson=1, Exponential=1, Gamma=2, Beta=2)[name]


def lik_param(name, **kw):
    if name == "Gaussian":
        s = kw.get("sigma")
        return 0.5 if s is None else float(s)          # gaussian.py:21-24
    if name == "Categorical":
        return float(kw["K"])
    return 0.0


class _PinnedBlock(object):
    """Owner of one hipHostMalloc allocation; freed when the last NumPy view goes away."""

    def __init__(self, nbytes):
        self.ptr = lib.hmogp_host_alloc(int(nbytes))
        if not self.ptr:
            raise MemoryError("hmogp_host_alloc(%d) failed" % nbytes)
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib.hmogp_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """NumPy array in page-locked host memory (hmogp_host_alloc): parameters kept in such arrays, and the gradients of an
    ``Engine(reuse_outputs=True)``, move between host and HBM without the driver's staging copy."""
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if shape else 1
    blk = _PinnedBlock(max(1, n) * dt.itemsize)
    buf = (C.c_char * blk.nbytes).from_address(blk.ptr)
    buf._owner = blk          # every NumPy view keeps `buf` (its base) alive, and with it the allocation
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


class Engine(object):
    """specs: list of (likelihood class name, kwargs) per task, e.g. [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {})]."""

    def __init__(self, specs, Q, M, P, device=0, chunk_rows=0, exact_zero_windows=False, cache_kuu=False, small_path=True,
                 reuse_outputs=False, quirks="reference", strict_qf=False):
        # strict_qf=True (HMOGP_CFG_STRICT_QF): q(f) and the row side of the gradients through the reference's solve-based forms
        # (svmogp_inf.py:214-218, :144-161) instead of the explicit C_q -- parity within 1e-5 also where GPy's jitter ladder is taken
        self.strict_qf = bool(strict_qf)
        self.specs = [(n, dict(k)) for n, k in specs]
        self.T, self.Q, self.M, self.P = len(specs), int(Q), int(M), int(P)
        f_index, d_index = [], []
        for t, (name, kw) in enumerate(self.specs):
            df = lik_dim_f(name, **kw)
            f_index += [t] * df
            d_index += list(range(df))
        self.f_index = np.array(f_index, dtype=np.int32)
        self.d_index = np.array(d_index, dtype=np.int32)
        self.Df = len(f_index)
        self.Mtri = self.M * (self.M + 1) // 2
        lik_id = np.array([LIK_IDS[n] for n, _ in self.specs], dtype=np.int32)
        lik_par = np.array([lik_param(n, **k) for n, k in self.specs], dtype=np.float64)
        cfg = _lib.Config(_lib.ABI_VERSION, self.T, self.Q, self.M, self.P, self.Df,
                          lik_id.ctypes.data_as(_lib.c_int32_p), _p(lik_par),
                          self.f_index.ctypes.data_as(_lib.c_int32_p), self.d_index.ctypes.data_as(_lib.c_int32_p),
                          int(device), int(chunk_rows),
                          (_lib.CFG_EXACT_ZERO_WINDOWS if exact_zero_windows else 0) | (_lib.CFG_CACHE_KUU if cache_kuu else 0) |
                          (0 if small_path else _lib.CFG_NO_SMALL_PATH) | (_lib.CFG_STRICT_QF if strict_qf else 0),
                          _lib.quirk_mask(quirks))
        self.quirks = _lib.quirk_mask(quirks)
        self._h = C.c_void_p()
        check(lib.hmogp_create(C.byref(cfg), C.byref(self._h)), None)
        self.N = [0] * self.T
        self.last = None
        # reuse_outputs=True: g_m_u / g_L_u / g_Z are returned as views of page-locked arrays owned by the engine (DMA
        # without the driver's staging copy), and the small outputs as engine-owned arrays too (the hmogp_outputs struct is
        # built once: ~25 us of ctypes work per call otherwise); ALL are overwritten by the next evaluation -- copy what
        # must persist.
        self.reuse_outputs = bool(reuse_outputs)
        self._pcache = None
        self._ocache = {}
        self._pinned_out = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib.hmogp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------ data
    def set_task_data(self, t, X, Y):
        X, Y = _f64(X).reshape(-1, self.P), _f64(Y).reshape(-1)
        if X.shape[0] != Y.shape[0]:
            raise ValueError("X and Y of task %d have different lengths" % t)
        check(lib.hmogp_set_task_data(self._h, t, _p(X), _p(Y), X.shape[0]), self._h)
        self.N[t] = X.shape[0]

    def set_data(self, X_list, Y_list):
        for t, (X, Y) in enumerate(zip(X_list, Y_list)):
            self.set_task_data(t, X, Y)

    # ------------------------------------------------------------------------------------------ params
    def _params(self, Z, m_u, L_flat, variance, lengthscale, W, kappa, W0=None, kappa0=None, batch_scale=None,
                row_begin=None, row_end=None, forced_rung=None, group_mask=_lib.GROUP_ALL, strict_qf=None, skip_g_L=False):
        """hmogp_params for one call.  The struct and the addresses of the arrays it points to are kept between calls: an
        argument that is THE SAME C-contiguous array object as last time (an optimiser updating its parameters in place) costs
        one identity check; anything else is converted (copied if it has to be) and its address taken again."""
        Q, M, P, Df, T = self.Q, self.M, self.P, self.Df, self.T
        if self._pcache is None:
            f8, i8, i4 = np.dtype(np.float64), np.dtype(np.int64), np.dtype(np.int32)
            spec = dict(Z=(M * Q * P, f8), m_u=(M * Q, f8), L_flat=(self.Mtri * Q, f8), variance=(Q, f8), lengthscale=(Q, f8),
                        W=(Q * Df, f8), kappa=(Q * Df, f8), W0=(Q * Df, f8), kappa0=(Q * Df, f8), batch_scale=(T, f8),
                        row_begin=(T, i8), row_end=(T, i8), forced_rung=(Q, i4))
            self._pcache = (_lib.Params(), spec, {k: None for k in spec}, {})
        p, spec, last, keep = self._pcache
        for k, a in (("Z", Z), ("m_u", m_u), ("L_flat", L_flat), ("variance", variance), ("lengthscale", lengthscale), ("W", W),
                     ("kappa", kappa), ("W0", W0), ("kappa0", kappa0), ("batch_scale", batch_scale), ("row_begin", row_begin),
                     ("row_end", row_end), ("forced_rung", forced_rung)):
            if a is last[k] and a is not None:
                continue
            if a is None:
                setattr(p, k, None)
                last[k] = keep[k] = None
                continue
            size, dt = spec[k]
            b = a if (type(a) is np.ndarray and a.dtype == dt and a.flags.c_contiguous) else np.ascontiguousarray(a, dtype=dt)
            if b.size != size:
                raise ValueError("%s: expected %d elements, got an array of shape %s" % (k, size, np.shape(a)))
            setattr(p, k, b.ctypes.data)
            keep[k] = b
            last[k] = a if b is a else None      # (a converted copy does not follow later in-place changes of `a`: convert again)
        p.group_mask = int(group_mask)
        p.eval_flags = (_lib.EVAL_STRICT_QF if strict_qf else 0) | (_lib.EVAL_NO_G_L if skip_g_L else 0)   # (per evaluation)
        return p, keep

    def _outputs(self, want_dL_dS=False, skip_qu=False):
        Q, M, P, Df = self.Q, self.M, self.P, self.Df
        key = (bool(want_dL_dS), bool(skip_qu))
        if self.reuse_outputs and key in self._ocache:     # every output array is owned by the engine and overwritten by the
            c, o = self._ocache[key]                        # next evaluation; the struct pointing at them is built once
            return c, dict(o)
        if self.reuse_outputs:      # page-locked arrays owned by the engine, overwritten by the next evaluation
            if self._pinned_out is None:
                self._pinned_out = dict(g_m_u=pinned_empty((M, Q)), g_L_u=pinned_empty((self.Mtri, Q)),
                                        g_Z=pinned_empty((M, Q * P)))
            big = self._pinned_out
        else:
            big = dict(g_m_u=np.zeros((M, Q)), g_L_u=np.zeros((self.Mtri, Q)), g_Z=np.zeros((M, Q * P)))
        o = dict(elbo=np.zeros(1), g_m_u=big["g_m_u"], g_L_u=big["g_L_u"], g_variance=np.zeros(Q),
                 g_lengthscale=np.zeros(Q), g_W=np.zeros((Q, Df)), g_kappa=np.zeros((Q, Df)), g_Z=big["g_Z"], kl=np.zeros(Q),
                 cond_est=np.zeros(Q))
        if want_dL_dS:
            o["dL_dS"] = np.zeros((Q, M, M))
        rung = np.zeros(Q, dtype=np.int32)
        flags = np.zeros(1, dtype=np.uint32)
        c = _lib.Outputs()
        for k, v in o.items():
            setattr(c, k, v.ctypes.data)
        if not want_dL_dS:
            c.dL_dS = None
        if skip_qu:                  # q(u) is device-resident: its gradient stays in HBM for hmogp_qu_adadelta
            c.g_m_u = None
            c.g_L_u = None
            o["g_m_u"] = o["g_L_u"] = None
        c.rung = rung.ctypes.data
        c.flags = flags.ctypes.data
        o["rung"], o["flags"] = rung, flags
        if self.reuse_outputs:
            self._ocache[key] = (c, dict(o))
        return c, o

    def _wrap(self, o):
        res = dict(o)
        res["elbo"] = float(o["elbo"][0])
        res["KL"] = float(np.sum(o["kl"]))          # calculate_KL (svmogp_inf.py:227-250), summed over the latents
        res["rungs"] = [int(r) for r in o["rung"]]
        res["v_negative"] = bool(int(o["flags"][0]) & _lib.FLAG_V_NEGATIVE)
        # cond_est[q] = variance_q max_i (K_uu^-1)_ii (lower bound of cond(K_uu), 30-150x below it); ill_conditioned: beyond what this
        # engine's mode keeps within element-wise 1e-5 of the reference (default mode: take strict_qf=True)
        res["ill_conditioned"] = bool(int(o["flags"][0]) & _lib.FLAG_ILL_CONDITIONED)
        self.last = res
        return res

    # ------------------------------------------------------------------------------------------ hot path
    def elbo_grad(self, want_dL_dS=False, sharded=False, **params):
        """One ``parameters_changed()``: returns dict(elbo, KL, kl [Q], g_m_u, g_L_u, g_variance, g_lengthscale, g_W,
        g_kappa, g_Z, rungs, v_negative[, dL_dS]).  `sharded=True` = hmogp_elbo_grad_sharded: the row-sharded step of a
        multi-GPU run (begin on this rank's rows -> in-library all-reduce -> finish), COLLECTIVE over the ranks of the
        communicator attached with `comm_init`; the default never communicates.
        With `Engine(reuse_outputs=True)` EVERY returned array (the three large gradients AND the small ones: g_variance, g_W,
        kl, rung, flags ...) is owned by the engine and overwritten in place by the next evaluation -- `engine.last` and any
        retained result dict change with it; copy what must persist."""
        p, keep = self._params(**params)
        c, o = self._outputs(want_dL_dS, skip_qu=params.get("m_u") is None)
        fn = lib.hmogp_elbo_grad_sharded if sharded else lib.hmogp_elbo_grad
        check(fn(self._h, C.byref(p), C.byref(c)), self._h)
        return self._wrap(o)

    def step_begin(self, **params):
        self._skip_qu = params.get("m_u") is None      # device-resident q(u): step_finish leaves its gradient in HBM too
        p, keep = self._params(**params)
        check(lib.hmogp_step_begin(self._h, C.byref(p)), self._h)

    def stats_buffer(self):
        """(device pointer, float64 count) of the additive statistic bundle -- what a multi-GPU run all-reduces."""
        ptr, n = C.c_void_p(), C.c_int64()
        check(lib.hmogp_stats_buffer(self._h, C.byref(ptr), C.byref(n)), self._h)
        return ptr.value, n.value

    def stats_read(self):
        _, n = self.stats_buffer()
        out = np.zeros(n)
        check(lib.hmogp_stats_read(self._h, _p(out)), self._h)
        return out

    def stats_write(self, host):
        host = _f64(host)
        check(lib.hmogp_stats_write(self._h, _p(host)), self._h)

    def wire_buffer(self):
        """(device pointer, float64 count) of the wire format of the bundle (lower triangles of H_q only)."""
        ptr, n = C.c_void_p(), C.c_int64()
        check(lib.hmogp_wire_buffer(self._h, C.byref(ptr), C.byref(n)), self._h)
        return ptr.value, n.value

    def wire_pack(self):
        check(lib.hmogp_wire_pack(self._h), self._h)

    def wire_unpack(self):
        check(lib.hmogp_wire_unpack(self._h), self._h)

    def wire_read(self):
        _, n = self.wire_buffer()
        out = np.zeros(n)
        check(lib.hmogp_wire_read(self._h, _p(out)), self._h)
        return out

    def wire_write(self, host):
        host = _f64(host)
        check(lib.hmogp_wire_write(self._h, _p(host)), self._h)

    # ------------------------------------------------------------------------------------------ native exchange
    def comm_init(self, nranks, rank, unique_id):
        """Attach an RCCL communicator (collective: every rank, same `unique_id` from `comm_unique_id()` of ONE rank).
        From then on `elbo_grad(sharded=True)` is the row-sharded step: begin -> in-library all-reduce -> finish."""
        uid = bytes(unique_id)
        if len(uid) != _lib.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.COMM_ID_BYTES)
        buf = C.create_string_buffer(uid, _lib.COMM_ID_BYTES)
        check(lib.hmogp_comm_init(self._h, int(nranks), int(rank), buf), self._h)

    def comm_destroy(self):
        check(lib.hmogp_comm_destroy(self._h), self._h)

    def comm_info(self):
        n, r = C.c_int32(), C.c_int32()
        check(lib.hmogp_comm_info(self._h, C.byref(n), C.byref(r)), self._h)
        return n.value, r.value

    def step_exchange(self):
        """Enqueue pack -> RCCL all-reduce -> unpack on the engine's stream (between step_begin and step_finish)."""
        check(lib.hmogp_step_exchange(self._h), self._h)

    def step_finish(self, want_dL_dS=False):
        c, o = self._outputs(want_dL_dS, skip_qu=getattr(self, "_skip_qu", False))
        check(lib.hmogp_step_finish(self._h, C.byref(c)), self._h)
        return self._wrap(o)

    # ------------------------------------------------------------------------------------------ consumers
    def posterior_u(self):
        wv = np.zeros((self.Q, self.M))
        wi = np.zeros((self.Q, self.M, self.M))
        check(lib.hmogp_posterior_u(self._h, _p(wv), _p(wi)), self._h)
        return wv, wi

    def natgrad_step(self, gamma=1.0):
        """Natural-gradient update of q(u) from the last evaluation's gradients: returns (m_u_new, L_flat_new)."""
        m = np.zeros((self.M, self.Q))
        L = np.zeros((self.Mtri, self.Q))
        check(lib.hmogp_natgrad_step(self._h, float(gamma), _p(m), _p(L)), self._h)
        return m, L

    def predict_f(self, Xnew):
        Xnew = _f64(Xnew).reshape(-1, self.P)
        m = np.zeros((Xnew.shape[0], self.Df))
        v = np.zeros((Xnew.shape[0], self.Df))
        check(lib.hmogp_predict_f(self._h, _p(Xnew), Xnew.shape[0], _p(m), _p(v)), self._h)
        return m, v

    # ------------------------------------------------------------------------------------------ resident q(u)
    def qu_load(self, m_u, L_flat):
        m_u, L_flat = _f64(m_u).reshape(self.M, self.Q), _f64(L_flat).reshape(self.Mtri, self.Q)
        check(lib.hmogp_qu_load(self._h, _p(m_u), _p(L_flat)), self._h)

    def qu_read(self):
        m, L = np.zeros((self.M, self.Q)), np.zeros((self.Mtri, self.Q))
        check(lib.hmogp_qu_read(self._h, _p(m), _p(L)), self._h)
        return m, L

    def qu_natgrad(self, gamma=1.0):
        """Natural-gradient step on the DEVICE-RESIDENT q(u) (hmogp_qu_natgrad): in place in HBM, from the gradients of the
        last evaluation.  Raises LinAlgError (HMOGP_E_NOT_PD) with q(u) untouched when gamma is too large."""
        check(lib.hmogp_qu_natgrad(self._h, float(gamma)), self._h)

    def qu_natgrad_async(self, gamma=1.0):
        """hmogp_qu_natgrad_async (ABI v7): the same step without the host synchronisation -- committed on the device only if it
        stays inside the positive-definite cone; `qu_natgrad_status()` tells later.  The next evaluation may be enqueued at once."""
        check(lib.hmogp_qu_natgrad_async(self._h, float(gamma)), self._h)

    def qu_natgrad_status(self):
        """True if the last asynchronous natural-gradient step was committed (waits for it if it is still pending)."""
        t = C.c_int32(0)
        check(lib.hmogp_qu_natgrad_status(self._h, C.byref(t)), self._h)
        return bool(t.value)

    def qu_adadelta(self, phase, step_rate, momentum, decay, offset):
        check(lib.hmogp_qu_adadelta(self._h, int(phase), float(step_rate), float(momentum), float(decay), float(1 - decay),
                                    float(offset)), self._h)

    def debug_raw_grads(self, rows):
        """The reference's inner-protocol gradient dict of the last evaluation (small N, one pool, GROUP_ALL):
        rows[t] = rows of task t in that evaluation.  Returns dict(dL_dKmm [Q][M,M], dL_dKmn [Q][Df][M,N_t],
        dL_dKdiag [Q][Df][N_t])."""
        Q, M, Df = self.Q, self.M, self.Df
        nd = [int(rows[self.f_index[d]]) for d in range(Df)]
        kmm = np.zeros((Q, M, M))
        kmn = np.zeros(Q * M * sum(nd))
        kdg = np.zeros(Q * sum(nd))
        check(lib.hmogp_debug_raw_grads(self._h, _p(kmm), _p(kmn), _p(kdg)), self._h)
        out = dict(dL_dKmm=[kmm[q] for q in range(Q)], dL_dKmn=[], dL_dKdiag=[])
        o1 = o2 = 0
        for q in range(Q):
            a, b = [], []
            for d in range(Df):
                a.append(kmn[o1:o1 + M * nd[d]].reshape(M, nd[d]))
                b.append(kdg[o2:o2 + nd[d]].copy())
                o1 += M * nd[d]
                o2 += nd[d]
            out["dL_dKmn"].append(a)
            out["dL_dKdiag"].append(b)
        return out

    def graph_stats(self):
        """(graphs captured, evaluations replayed) of the small-model hipGraph mechanism (hmogp_graph_stats)."""
        cap, rep = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64)
        check(lib.hmogp_graph_stats(self._h, cap.ctypes.data_as(_lib.c_int64_p), rep.ctypes.data_as(_lib.c_int64_p)), self._h)
        return int(cap[0]), int(rep[0])

    def timings(self):
        ms = np.zeros(_lib.NTIMINGS)
        n = np.zeros(_lib.NTIMINGS, dtype=np.int64)
        check(lib.hmogp_last_timings(self._h, _p(ms), n.ctypes.data_as(_lib.c_int64_p)), self._h)
        names = ["total", "rbf_cross_cov", "forward_gemm", "rowstats_combine", "quadrature", "gram_gemm", "colstats_reduce",
                 "mxm_algebra", "exchange", "trsm_solves", "strict_rowstats"]   # (the last two: strict q(f) mode only, ABI v8)
        return dict(zip(names, ms.tolist())), dict(zip(names, n.tolist()))


_DEFAULT_DEVICE = 0


_DEFAULT_DEVICE_PINNED = False


def set_default_device(device, from_model=False):
    """HIP device of the stand-alone building blocks below (var_exp, predictive, sample, gemm ...) when they are called
    without `device=`: a rank of a multi-GPU run sets it to its LOCAL_RANK once (SVMOGP(distributed=True) does), so that
    likelihood-level helpers never land on GPU 0 from every rank.  A model constructor (`from_model=True`) only sets
    it while nobody has chosen one explicitly and no earlier model has: with two models on two devices in one process the
    stand-alone helpers of the first no longer move to the GPU of the last; a model's OWN calls always pass its device."""
    global _DEFAULT_DEVICE, _DEFAULT_DEVICE_PINNED
    if from_model and _DEFAULT_DEVICE_PINNED:
        return
    _DEFAULT_DEVICE = int(device)
    _DEFAULT_DEVICE_PINNED = True


def _resolve_device(device):
    return _DEFAULT_DEVICE if device is None else int(device)


def comm_available():
    """True if the library can load librccl (needed only for multi-GPU runs)."""
    return bool(lib.hmogp_comm_available())


def comm_unique_id():
    """ncclGetUniqueId through the library: call on ONE rank, broadcast the bytes, pass them to Engine.comm_init."""
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    check(lib.hmogp_comm_unique_id(buf))
    return buf.raw


# ---------------------------------------------------------------------------------------------- building blocks
def rbf_cross_cov(X, Z, variance, lengthscale, device=None, exact=True):
    """K = k(X, Z).  exact=True: GPy's rounding order (the K_uu variant); exact=False: the hot-path K_uf variant."""
    device = _resolve_device(device)
    X, Z = _f64(X), _f64(Z)
    X = X.reshape(X.shape[0], -1)
    Z = Z.reshape(Z.shape[0], -1)
    K = np.zeros((X.shape[0], Z.shape[0]))
    check(lib.hmogp_rbf_cross_cov_ex(device, _p(X), X.shape[0], _p(Z), Z.shape[0], X.shape[1], float(variance),
                                     float(lengthscale), 1 if exact else 0, _p(K)))
    return K


def jitchol_inv(A, forced_rung=None, device=None):
    device = _resolve_device(device)
    A = _f64(A)
    Q, M = A.shape[0], A.shape[1]
    L, Ai = np.zeros_like(A), np.zeros_like(A)
    rung = np.zeros(Q, dtype=np.int32)
    fr = None if forced_rung is None else np.ascontiguousarray(forced_rung, dtype=np.int32)
    check(lib.hmogp_jitchol_inv(device, _p(A), Q, M, fr.ctypes.data_as(_lib.c_int32_p) if fr is not None else None,
                                _p(L), _p(Ai), rung.ctypes.data_as(_lib.c_int32_p)))
    return L, Ai, [int(r) for r in rung]


def potri(L, device=None):
    device = _resolve_device(device)
    L = _f64(L)
    out = np.zeros_like(L)
    check(lib.hmogp_potri(device, _p(L), L.shape[0], L.shape[1], _p(out)))
    return out


def potrs_rows(L, B, device=None):
    """B (L L^T)^-1 for the rows of B [n, M]: GPy's dpotrs(L, B^T)^T (svmogp_inf.py:214) by blocked substitution on the device."""
    device = _resolve_device(device)
    L, B = _f64(L), _f64(B)
    out = np.zeros_like(B)
    check(lib.hmogp_potrs_rows(device, _p(L), L.shape[0], _p(B), B.shape[0], _p(out)))
    return out


def gemm(A, B, transA=False, transB=False, alpha=1.0, beta=0.0, C0=None, device=None):
    device = _resolve_device(device)
    A, B = _f64(A), _f64(B)
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    N = B.shape[0] if transB else B.shape[1]
    Cm = np.zeros((M, N)) if C0 is None else _f64(C0).copy()
    check(lib.hmogp_gemm_f64(device, int(transA), int(transB), M, N, K, float(alpha), _p(A), A.shape[1], _p(B), B.shape[1],
                             float(beta), _p(Cm), N))
    return Cm


def var_exp(name, y, m, v, device=None, quirks="reference", **kw):
    device = _resolve_device(device)
    y, m, v = _f64(y).reshape(-1), _f64(m), _f64(v)
    J = lik_dim_f(name, **kw)
    m, v = m.reshape(-1, J), v.reshape(-1, J)
    ve, dm, dv = np.zeros(y.shape[0]), np.zeros_like(m), np.zeros_like(v)
    check(lib.hmogp_var_exp_ex(device, LIK_IDS[name], lik_param(name, **kw), _lib.quirk_mask(quirks), y.shape[0], _p(y),
                               _p(m), _p(v), _p(ve), _p(dm), _p(dv)))
    return ve, dm, dv


def predictive(name, m, v, gh_T=0, device=None, **kw):
    """`<likelihood>.predictive(m, v)`: predictive mean / variance of y (N, dim_p)."""
    device = _resolve_device(device)
    m, v = _f64(m), _f64(v)
    J = lik_dim_f(name, **kw)
    m, v = m.reshape(-1, J), v.reshape(-1, J)
    Jp = J if name == "Categorical" else 1
    mean, var = np.zeros((m.shape[0], Jp)), np.zeros((m.shape[0], Jp))
    check(lib.hmogp_predictive(device, LIK_IDS[name], lik_param(name, **kw), int(gh_T), m.shape[0], _p(m), _p(v), _p(mean),
                               _p(var)))
    return mean, var


def log_predictive_rows(name, y, m, v, num_samples=1000, seed=0, device=None, **kw):
    """Per-row Monte-Carlo log predictive density: -log S + logsumexp_s log p(y_n | f_s), f_s ~ N(m_n, diag v_n)."""
    device = _resolve_device(device)
    y, m, v = _f64(y).reshape(-1), _f64(m), _f64(v)
    J = lik_dim_f(name, **kw)
    m, v = m.reshape(-1, J), v.reshape(-1, J)
    out = np.zeros(y.shape[0])
    check(lib.hmogp_log_predictive(device, LIK_IDS[name], lik_param(name, **kw), y.shape[0], int(num_samples), int(seed),
                                   _p(y), _p(m), _p(v), _p(out)))
    return out


def sample(name, F, seed=0, device=None, **kw):
    """One draw y ~ p(y | F[n]) per row on the device (the reference's `<likelihood>.samples`): returns (N, 1)."""
    device = _resolve_device(device)
    J = lik_dim_f(name, **kw)
    F = _f64(F).reshape(-1, J)
    Y = np.zeros(F.shape[0])
    check(lib.hmogp_sample(device, LIK_IDS[name], lik_param(name, **kw), F.shape[0], int(seed) & (2 ** 64 - 1), _p(F), _p(Y)))
    return Y[:, None]
