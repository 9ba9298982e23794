"""GPU: the strict q(f) mode (HMOGP_CFG_STRICT_QF) beyond the three jitter-ladder fixtures of tests/test_gpu_ladder.py --
every dispatch branch the mode adds, against the reference-run fixtures and against the oracle's strict restatement:

  * every reference-run fixture of the repository (inf_* / model_* / ref_*: 1-D and 2-D inputs, all eight likelihoods, ragged M = 5 ...
    160, minibatch E- and M-steps with batch scales, stale-W quirk) must pass in strict mode under the same criterion as in default mode;
  * seeded random configurations (ragged / 32-multiple / 128-multiple M, P = 1, 2, several pools, minibatch row ranges, gradient gates)
    against `so.elbo_grad_fused(strict_qf=True)`;
  * additivity of the strict bundle over row shards (its H / r slots hold A^T diag(beta) A and A^T alpha: still sums over rows),
    the inner-protocol debug export, hmogp_potrs_rows against LAPACK, and the flag's exclusions."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity
from test_gpu_engine import KEYS, rel, run, synth

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "model_*.npz")) + glob.glob(os.path.join(GOLDEN, "ref_*.npz")))


def _engine(prob, X, Y, **kw):
    from hetmogp_amd.engine import Engine
    e = Engine(prob["specs"], prob["Q"], prob["M"], prob["P"], strict_qf=True, **kw)
    e.set_data(X, Y)
    return e


@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_strict_mode_vs_every_reference_run_fixture(path):
    from oracle import svmogp_oracle as so
    from hetmogp_amd import _lib
    g = np.load(path)
    prm, prob, X, Y, bs = so.load_case(g)
    mask = _lib.GROUP_ALL
    if bool(g["stochastic"]):
        mask = _lib.GROUP_QU if bool(g["vem_step"]) else (_lib.GROUP_HYPER | _lib.GROUP_Z)
    e = _engine(prob, X, Y)
    out = run(e, prm, bs, group_mask=mask)
    assert out["rungs"] == [-1] * prob["Q"]
    for k in KEYS:
        assert_parity(out[k], g[k], k)
    e.close()


LIKS = [("Gaussian", {"sigma": 0.7}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {}), ("Exponential", {}), ("HetGaussian", {}),
        ("Beta", {}), ("Categorical", {"K": 3})]


STRICT_MS = [16, 33, 50, 64, 96, 100, 128, 160, 256, 272, 384, 50, 144, 256]


@pytest.mark.parametrize("seed", range(14))
def test_strict_random_configuration_vs_strict_oracle(seed):
    _strict_case(seed)


def _strict_case(seed):
    """(seeds beyond the suite's 14 cycle through the same inducing counts: tools/soak_parity.py)"""
    from oracle import svmogp_oracle as so
    from hetmogp_amd import _lib
    rng = np.random.RandomState(5000 + seed)
    M = STRICT_MS[seed % len(STRICT_MS)]
    P = 2 if (seed % len(STRICT_MS)) >= 11 else 1
    Q = int(rng.randint(1, 4))
    T = int(rng.randint(1, 4))
    specs = [LIKS[i] for i in rng.choice(len(LIKS), T, replace=False)]
    Ns = [int(rng.choice([1, 17, 130, 257, 700])) for _ in range(T)]
    prm, prob, X, Y = synth(6000 + seed, specs, Ns, M, Q, P, tuple(0.9 + 0.4 * rng.rand(Q)))
    sprob = dict(prob, strict_qf=True)
    e = _engine(prob, X, Y, chunk_rows=int(rng.choice([1 << 20, 300])))
    want = so.elbo_grad_fused(prm, sprob, X, Y)
    out = run(e, prm)
    for k in KEYS:
        assert rel(out[k], want[k]) < 1e-8, (k, M, P, Q, specs, Ns, rel(out[k], want[k]))
    # minibatch row ranges with batch scales; E-step gate (q(u) group only: the mode then skips P~ and the gradient statistics)
    rb = [int(rng.randint(0, n // 2 + 1)) for n in Ns]
    re = [int(min(n, b + max(1, n // 3))) for n, b in zip(Ns, rb)]
    bs = [float(n) / max(e_ - b, 1) for n, b, e_ in zip(Ns, rb, re)]
    Xs, Ys = [x[b:e_] for x, b, e_ in zip(X, rb, re)], [y[b:e_] for y, b, e_ in zip(Y, rb, re)]
    wantb = so.elbo_grad_fused(prm, sprob, Xs, Ys, batch_scale=bs, stochastic=True, vem_step=True)
    outb = run(e, prm, bs, row_begin=rb, row_end=re, group_mask=_lib.GROUP_QU)
    for k in ("elbo", "g_m_u", "g_L_u"):
        assert rel(outb[k], wantb[k]) < 1e-8, ("E-step", k, rel(outb[k], wantb[k]))
    assert not np.any(outb["g_Z"]) and not np.any(outb["g_W"])
    e.close()


@pytest.mark.parametrize("M,Ns", [(256, [3000, 2600]), (512, [2500, 2200, 1900]), (384, [4100, 300])])
def test_strict_solves_through_the_specialised_update_kernel(M, Ns):
    """>= 4096 rows and M a multiple of 128: the 128-column updates of the two triangular solves run through rowpass_gemm_kernel<1>
    (C -= A B, mirrored factor for the forward solve), every column block reaches A by its first touch (no copy of K^), the in-block
    updates ride in the substitution launches and rowsum(T^2) comes out of the fold kernel -- against the oracle's strict
    restatement, full gradients and an E-step."""
    from oracle import svmogp_oracle as so
    from hetmogp_amd import _lib
    specs = [("Gaussian", {"sigma": 0.5}), ("Poisson", {}), ("Bernoulli", {})][:len(Ns)]
    prm, prob, X, Y = synth(7000 + M, specs, Ns, M, 2, 1, (1.0, 1.2))
    sprob = dict(prob, strict_qf=True)
    want = so.elbo_grad_fused(prm, sprob, X, Y)
    e = _engine(prob, X, Y)
    out = run(e, prm)
    for k in KEYS:
        assert rel(out[k], want[k]) < 1e-8, (k, M, rel(out[k], want[k]))
    qu = run(e, prm, group_mask=_lib.GROUP_QU)
    for k in ("elbo", "g_m_u", "g_L_u"):
        assert rel(qu[k], want[k]) < 1e-8, ("E-step", k, rel(qu[k], want[k]))
    e.close()


@pytest.mark.timeout(600)
def test_strict_mode_at_the_headline_M_in_several_pools():
    """M = 1024 (32 diagonal blocks per triangular solve, 128-column MFMA tiles for the Gram of A), Q = 3, the headline mix, rows
    streamed in three pools that cross task boundaries: against the oracle's strict restatement, and equal to the single-pool run."""
    from oracle import svmogp_oracle as so
    from hetmogp_amd.synthetic import make_case
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
    M, Q = 1024, 3
    prm, X, Y = make_case(specs, [2500, 2300, 2700, 2100], M=M, Q=Q, P=1, seed=20260936)
    prob = so.make_problem(specs, Q, M, 1)
    want = so.elbo_grad_fused(prm, dict(prob, strict_qf=True), X, Y)
    e1, e3 = _engine(prob, X, Y), _engine(prob, X, Y, chunk_rows=3500)
    a, b = e1.elbo_grad(**prm), e3.elbo_grad(**prm)
    for k in KEYS:
        assert rel(a[k], want[k]) < 1e-8, (k, rel(a[k], want[k]))
        assert rel(b[k], a[k]) < 1e-10, ("pools", k, rel(b[k], a[k]))
    # [r6] an E-step at this M: the one-solve form end to end through the one-launch-per-block kernels -- the forward solve of the
    # 9600-row pool with the fused statistics, the stacked 2 M + 1-row solve of the right factors, and `finish`'s two backward
    # solves of M + 1 and M rows (ragged last row tile) -- against the oracle's one-solve restatement AND its two-solve one
    from hetmogp_amd import _lib
    w1 = so.elbo_grad_fused(prm, dict(prob, strict_qf="one_solve"), X, Y, stochastic=True, vem_step=True)
    w2 = so.elbo_grad_fused(prm, dict(prob, strict_qf=True), X, Y, stochastic=True, vem_step=True)
    qu = e1.elbo_grad(group_mask=_lib.GROUP_QU, **prm)
    ms = e1.timings()[0]
    assert ms["trsm_solves"] > 0.0 and ms["strict_rowstats"] >= 0.0          # (ABI v8 timing slots)
    for k in ("elbo", "g_m_u", "g_L_u"):
        assert rel(qu[k], w1[k]) < 1e-8, ("E-step vs one-solve oracle", k, rel(qu[k], w1[k]))
        assert rel(qu[k], w2[k]) < 1e-8, ("E-step vs two-solve oracle", k, rel(qu[k], w2[k]))
    e1.close(), e3.close()


def test_strict_bundle_is_additive_over_row_shards_and_debug_export_matches():
    from oracle import svmogp_oracle as so
    g = np.load(os.path.join(GOLDEN, "ref_h_mix_M128.npz"))
    prm, prob, X, Y, bs = so.load_case(g)
    prm.pop("W0", None)
    e = _engine(prob, X, Y)
    full = e.elbo_grad(batch_scale=bs, **prm)
    T = prob["T"]
    cut = [x.shape[0] // 3 + 7 * t for t, x in enumerate(X)]
    e.step_begin(row_begin=[0] * T, row_end=cut, batch_scale=bs, **prm)
    s1 = e.stats_read()
    e.step_begin(row_begin=cut, row_end=[x.shape[0] for x in X], batch_scale=bs, **prm)
    s2 = e.stats_read()
    e.stats_write(s1 + s2)
    both = e.step_finish()
    for k in KEYS:
        assert rel(both[k], full[k]) < 1e-10, ("shards", k, rel(both[k], full[k]))
    # inner protocol (svmogp_inf.py:107) in strict mode against the reference's own dict
    e.elbo_grad(batch_scale=bs, **prm)
    raw = e.debug_raw_grads([x.shape[0] for x in X])
    for q in range(prob["Q"]):
        assert_parity(raw["dL_dKmm"][q], g["dL_dKmm_%d" % q], ("dL_dKmm", q))
        for d in range(prob["Df"]):
            assert_parity(np.sum(raw["dL_dKmn"][q][d], axis=1), g["dL_dKmn_rowsum_%d_%d" % (q, d)], ("dL_dKmn", q, d))
    e.close()


@pytest.mark.parametrize("M", [8, 31, 32, 33, 64, 100, 128, 257])
def test_potrs_rows_vs_lapack(M):
    """hmogp_potrs_rows == dpotrs(L, B^T)^T (svmogp_inf.py:214): blocked substitution, ragged last block, well- and ill-conditioned."""
    import scipy.linalg as sl
    from hetmogp_amd.engine import potrs_rows
    rng = np.random.RandomState(M)
    A = rng.randn(M, M)
    for jitter in (float(M), 1e-6):
        K = A @ A.T / M + jitter * np.eye(M)
        L = np.linalg.cholesky(K)
        B = rng.randn(333, M)
        ref = sl.cho_solve((L, True), B.T).T
        out = potrs_rows(L, B)
        bound = 50.0 * np.linalg.cond(K) * 2.2e-16
        assert np.max(np.abs(out - ref)) <= max(1e-13, bound) * np.max(np.abs(ref)), (M, jitter)


@pytest.mark.parametrize("M,n", [(128, 1024), (256, 1500), (384, 2049), (1024, 1153)])
def test_potrs_rows_panel_kernels_vs_lapack(M, n):
    """[r6] hmogp_potrs_rows at shapes the one-launch-per-block kernels take (trsm_panel.hip: M a multiple of 128, >= 1024 rows, ragged
    last row tile): long-K update + in-register 4-column substitution groups against LAPACK's dpotrs, well- and ill-conditioned; and
    row by row equal to the same call on a 333-row slice (which takes the round-5 kernels): two valid blocked substitutions."""
    import scipy.linalg as sl
    from hetmogp_amd.engine import potrs_rows
    rng = np.random.RandomState(M + n)
    A = rng.randn(M, M)
    for jitter in (float(M), 1e-6):
        K = A @ A.T / M + jitter * np.eye(M)
        L = np.linalg.cholesky(K)
        B = rng.randn(n, M)
        ref = sl.cho_solve((L, True), B.T).T
        out = potrs_rows(L, B)
        bound = 50.0 * np.linalg.cond(K) * 2.2e-16
        assert np.max(np.abs(out - ref)) <= max(1e-13, bound) * np.max(np.abs(ref)), (M, n, jitter)
        small = potrs_rows(L, B[:333])
        assert np.max(np.abs(out[:333] - small)) <= max(1e-13, bound) * np.max(np.abs(ref)), (M, n, jitter, "panel vs round-5 kernels")


def test_strict_flag_exclusions():
    from hetmogp_amd.engine import Engine
    from hetmogp_amd._lib import InvalidArgument
    with pytest.raises(InvalidArgument):
        Engine([("Gaussian", {"sigma": 0.5})], 1, 128, 1, strict_qf=True, exact_zero_windows=True)


def test_unknown_eval_flag_bits_are_refused():
    """[r6] ADVICE r5: hmogp_params.eval_flags (ABI v6/v7) was masked for its two known bits; a caller that fills a v5-sized struct
    without zeroing the new field could switch the strict mode on, or get a zero-filled g_L_u, with no error.  Unknown bits now
    return HMOGP_E_INVALID like those of hmogp_config.flags / quirks; the two defined bits still pass."""
    import ctypes as C
    from hetmogp_amd import _lib
    from hetmogp_amd.engine import Engine
    from hetmogp_amd.synthetic import make_case
    specs = [("Gaussian", {"sigma": 0.5})]
    prm, X, Y = make_case(specs, [600], M=32, Q=1, P=1, seed=3)
    e = Engine(specs, 1, 32, 1)
    e.set_data(X, Y)
    good = e.elbo_grad(**prm)["elbo"]
    p, keep = e._params(**prm)
    c, o = e._outputs()
    for bad in (4, 0x80000000, 1 | 8):
        p.eval_flags = bad
        rc = _lib.lib.hmogp_elbo_grad(e._h, C.byref(p), C.byref(c))
        assert rc == _lib.E_INVALID, (bad, rc)
        assert b"eval_flags" in _lib.lib.hmogp_last_error(e._h)
    p.eval_flags = _lib.EVAL_STRICT_QF | _lib.EVAL_NO_G_L
    assert _lib.lib.hmogp_elbo_grad(e._h, C.byref(p), C.byref(c)) == 0
    assert abs(e.elbo_grad(**prm)["elbo"] - good) <= 1e-12 * abs(good)      # and the engine is still usable afterwards
    e.close()
