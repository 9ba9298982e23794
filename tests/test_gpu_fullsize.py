"""GPU: the BASELINE.json configurations at their FULL sizes (H: 4 x 200 000 rows, M = 1024; C2: the same at M = 512; the
per-rank share of C4: 8 x 125 000 rows, Q = 4, Df = 14 -- 4.1e9-element K^ / P~ workspaces), where the oracle cannot run
the whole evaluation, through size-independent properties:

  * shard additivity   the statistic bundles of two row shards add up to the bundle of all rows (what the multi-GPU
                       exchange relies on) -- gradients after `finish` agree with the single evaluation;
  * chunk invariance   streaming the rows in several pools gives the same results as one pool;
  * repeatability      two evaluations are bit-identical (deterministic reductions, no floating-point atomics);
  * oracle parity      on a window of rows taken from the END of every task (row_begin close to N_t: the offsets into the
                       row workspaces are exercised at their largest values) against the NumPy oracle on exactly those rows;
  * dense == exact-zero windows (1-D sorted inputs): the opt-in mode drops only products with exact zeros.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"]
H_SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
C4_SPECS = [("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {}), ("Gaussian", {"sigma": 0.5}),
            ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def check_config(specs, N, M, Q, window, seed, pools, exact_zero=True, oracle_tol=1e-8):
    from hetmogp_amd.engine import Engine
    from hetmogp_amd.synthetic import make_case
    from oracle import svmogp_oracle as so
    T = len(specs)
    prm, X, Y = make_case(specs, [N] * T, M=M, Q=Q, P=1, seed=seed)
    e = Engine(specs, Q, M, 1)
    e.set_data(X, Y)
    full = e.elbo_grad(**prm)
    assert np.isfinite(full["elbo"]) and not full["v_negative"] and full["rungs"] == [-1] * Q
    # repeatability: same bits
    again = e.elbo_grad(**prm)
    for k in KEYS:
        assert np.array_equal(np.asarray(full[k]), np.asarray(again[k])), k
    # shard additivity (uneven split, different per task)
    cut = [N // 3 + 1000 * t for t in range(T)]
    e.step_begin(row_begin=[0] * T, row_end=cut, **prm)
    s1 = e.stats_read()
    e.step_begin(row_begin=cut, row_end=[N] * T, **prm)
    s2 = e.stats_read()
    e.stats_write(s1 + s2)
    both = e.step_finish()
    for k in KEYS:
        assert rel(both[k], full[k]) < 1e-9, ("shards", k, rel(both[k], full[k]))
    # oracle parity on the LAST `window` rows of every task
    prob = so.make_problem(specs, Q, M, 1)
    want = so.elbo_grad_fused(prm, prob, [x[N - window:] for x in X], [y[N - window:] for y in Y])
    got = e.elbo_grad(row_begin=[N - window] * T, row_end=[N] * T, **prm)
    for k in KEYS:
        assert rel(got[k], want[k]) < oracle_tol, ("oracle window", k, rel(got[k], want[k]))
    e.close()
    # chunk invariance: `pools` pools instead of one
    total = N * T
    e2 = Engine(specs, Q, M, 1, chunk_rows=(total + pools - 1) // pools + 17)
    e2.set_data(X, Y)
    chunked = e2.elbo_grad(**prm)
    for k in KEYS:
        assert rel(chunked[k], full[k]) < 1e-9, ("pools", k, rel(chunked[k], full[k]))
    e2.close()
    if exact_zero:
        e3 = Engine(specs, Q, M, 1, exact_zero_windows=True)
        e3.set_data(X, Y)
        ez = e3.elbo_grad(**prm)
        for k in KEYS:
            assert rel(ez[k], full[k]) < 1e-9, ("exact-zero windows", k, rel(ez[k], full[k]))
        e3.close()


@pytest.mark.timeout(600)
def test_headline_H_full_size():
    """H: T=4 [Gaussian, Bernoulli, Poisson, Gamma], N_t = 200 000, M = 1024, Q = 3 (the bench workload)."""
    check_config(H_SPECS, 200000, 1024, 3, window=2000, seed=20260929, pools=3)


@pytest.mark.timeout(600)
def test_C2_full_size():
    """C2: the same mix at M = 512 (the Gram's lower tiles are 4 diagonal + 6 off-diagonal: the diagonal path weighs most)."""
    check_config(H_SPECS, 200000, 512, 3, window=2000, seed=20260931, pools=2)


@pytest.mark.timeout(900)
def test_C4_rank_share_full_size():
    """One rank's share of C4: 8 tasks x 125 000 rows, Q = 4, Df = 14 (HetGaussian, Categorical(5), Beta, ...): the K^ / P~
    workspaces hold 4 x 1e6 x 1024 = 4.1e9 elements each -- element indices beyond 2^32."""
    check_config(C4_SPECS, 125000, 1024, 4, window=1000, seed=20260933, pools=2, exact_zero=False)
