"""GPU: seeded random configurations against the oracle.  The shapes are drawn so that every dispatch branch of the
contraction kernels is hit: inducing counts that are / are not multiples of 16, 64 and 128 (specialised 8-wave row-pass
kernels, general kernel, 64 x 64-tile M x M products), odd and even tile counts (two-phase Gram tile order), one to
three latents, 1-D and 2-D inputs, ragged / tiny / empty tasks, minibatch ranges, several row pools, both quirk modes."""
import numpy as np
import pytest

from test_gpu_engine import KEYS, make_engine, rel, run, synth

pytestmark = pytest.mark.gpu

LIKS = [("Gaussian", {"sigma": 0.7}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {}), ("Exponential", {}), ("HetGaussian", {}),
        ("Beta", {}), ("Categorical", {"K": 3})]
MS = [16, 48, 64, 100, 128, 192, 256, 272, 320, 384, 448, 512, 640]


MS_LARGE = [768, 1024, 1280, 2048]      # multiples of 128: specialised row-pass kernels, look-ahead Cholesky over 24-64 panels


@pytest.mark.parametrize("seed", range(26))
def test_random_configuration_vs_oracle(seed):
    _case(seed, MS, False)


@pytest.mark.parametrize("seed", range(100, 104))
def test_random_configuration_large_M_vs_oracle(seed):
    """The same at the inducing counts of the BASELINE configurations (1-D inputs: with 2-D grids of 1000+ inducing points
    cond(K_uu) ~ 1e6 limits the agreement of the two implementations to ~1e-6 in the cancelling sums of g_variance)."""
    _case(seed, MS_LARGE, True)


def _case(seed, ms, one_d):
    from oracle import svmogp_oracle as so
    rng = np.random.RandomState(1000 + seed)
    M = ms[seed % len(ms)]
    P = 1 if (rng.rand() < 0.7 or one_d) else 2
    Q = int(rng.randint(1, 4))
    T = int(rng.randint(1, 4))
    specs = [LIKS[i] for i in rng.choice(len(LIKS), T, replace=False)]
    Ns = [int(rng.choice([0, 1, 17, 130, 257, 700, 1500], p=[0.05, 0.05, 0.1, 0.2, 0.2, 0.2, 0.2])) for _ in range(T)]
    if sum(Ns) == 0:
        Ns[0] = 129
    cs = tuple(0.9 + 0.4 * rng.rand(Q))
    prm, prob, X, Y = synth(2000 + seed, specs, Ns, M, Q, P, cs)
    quirks = "exact" if seed % 3 == 0 else "reference"     # true gradients / the reference's (quirks Q1-Q5)
    prob = dict(prob, quirks=quirks)
    want = so.elbo_grad_fused(prm, prob, X, Y)
    e = make_engine(prob, X, Y, chunk_rows=int(rng.choice([1 << 20, 600, 256])), quirks=quirks)
    out = run(e, prm)
    # 2-D grids of inducing points with jitter have cond(K_uu) up to ~1e6: parity there is conditioning-limited.  The
    # yardstick is the distance between the oracle's OWN two restatements -- the literal one (solves,
    # the reference's operation order) and the fused one (explicit inverses, the engine's algebra): the engine may be no
    # further from the fused restatement than 10x that, and never further than 2e-7.
    # [r5] 2e-8 (2e-7 until round 4) wherever the engine itself does NOT flag K_uu as ill-conditioned for the explicit-inverse path
    # (hmogp_outputs.cond_est <= 5e2, i.e. cond <~ 1e4-1e5); where it does (2-D grids with l = 1.3 spacings reach cond 3e5: two valid
    # K_uu^-1 then differ by cond * eps * M = 2e-8 and everything KL-dominated with them -- tools/soak_parity.py, seeds 1011 / 1058)
    # the old 2e-7 stands.
    tol = 1e-8 if P == 1 else (2e-7 if out["ill_conditioned"] else 2e-8)
    flagged = bool(out["ill_conditioned"]) and P > 1
    lit = so.elbo_grad_literal(prm, prob, X, Y) if (P > 1 and quirks == "reference" and min(Ns) > 0 and not flagged) else None
    # [r5] quirk Q10: GPy's expanded-form distances clip |x - z| <~ 1e-8 to exactly 0 and its gradient code drops entries
    # with r == 0; the oracle follows it, the default path keeps the (true) term E K (x - z) / l^2.  A data point that close to an
    # inducing point is a ~1 % event per random configuration (tools/soak_parity.py, seed 3117: |x - z| = 4.7e-9, ONE entry of g_Z
    # 2.8e-7 of its value away, everything else 1e-13): g_Z is then held to 1e-5 like the reference-run fixtures.
    near = min([float(np.min(np.max(np.abs(x[:, None, :] - prm["Z"][None, :, q * P:(q + 1) * P]), axis=2))) for x in X if len(x) for q in range(Q)] or [1.0])
    for k in KEYS:
        tk = tol if lit is None else min(tol, max(1e-8, 10.0 * rel(want[k], lit[k])))
        if flagged and k == "g_variance":      # (a difference of sums ~100x its size: 3e-7 at cond 3.6e5 in the 300-seed soak)
            tk = 5e-7
        if k == "g_Z" and near < 1e-7:
            tk = max(tk, 1e-5)
        assert rel(out[k], want[k]) < tk, (k, M, P, Q, specs, Ns, tk)
    # a minibatch: a random contiguous range of every task with the reference's batch scale (svmogp.py:101-105)
    rb = [int(rng.randint(0, n // 2 + 1)) for n in Ns]
    re = [int(min(n, b + max(1, n // 3))) if n else 0 for n, b in zip(Ns, rb)]
    bs = [float(n) / max(e_ - b, 1) for n, b, e_ in zip(Ns, rb, re)]
    Xs, Ys = [x[b:e_] for x, b, e_ in zip(X, rb, re)], [y[b:e_] for y, b, e_ in zip(Y, rb, re)]
    probs = dict(prob)
    wantb = so.elbo_grad_fused(prm, probs, Xs, Ys, batch_scale=bs)
    outb = run(e, prm, bs, row_begin=rb, row_end=re)
    for k in KEYS:
        tk = max(tol, 1e-5) if (k == "g_Z" and near < 1e-7) else tol
        if flagged and k == "g_variance":      # (the same allowance as for the full batch above: [r6] soak seeds 20135 / 20407, 2.4e-8 / 1.0e-7
            tk = 5e-7                          #  on the full batch and more on a third of the rows with 3x the batch scale)
        assert rel(outb[k], wantb[k]) < tk, ("minibatch", k, M, P, Q, specs, Ns, rb, re)
