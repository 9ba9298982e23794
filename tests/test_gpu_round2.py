"""GPU: parity gaps closed in round 2 (VERDICT r1 "next round" items 4 and 5), all through the C ABI:
  * the HOT-PATH K_uf kernel variant (rbf_kernel<P, false>) directly against the oracle,
  * the inner-protocol debug export (dL_dKmm / dL_dKmn / dL_dKdiag) against the reference's own raw gradient dict,
  * a jitter ladder that is actually taken at M = 512,
  * the wire format of the exchange step (lower triangles of H_q),
  * quirks = "exact": finite differences of the ELBO in every parameter group, agreement with the oracle's exact mode."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_engine import KEYS, make_engine, rel, run, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,N,M", [(1, 100, 37), (1, 257, 64), (2, 65, 50), (3, 33, 16), (4, 70, 130), (1, 5, 1030)])
def test_rbf_cross_cov_hot_path_variant(P, N, M):
    """rbf_kernel<P, false> -- the variant the row pass launches for K_uf (engine.hip: kuf_pool) -- against GPy's RBF.K
    restated in the oracle.  It scales clip(r2) by 1/l^2 instead of forming sqrt(clip(r2))/l: <= 2 ulp of the exponent."""
    from hetmogp_amd import engine as E
    from oracle import svmogp_oracle as so
    rng = np.random.RandomState(P * 100 + N + M)
    X, Z = rng.rand(N, P), rng.rand(M, P)
    ell = 0.9 * M ** (-1.0 / P)
    want = so.rbf_K(X, Z, 0.7, ell)
    fast = E.rbf_cross_cov(X, Z, 0.7, ell, exact=False)
    exact = E.rbf_cross_cov(X, Z, 0.7, ell, exact=True)
    np.testing.assert_allclose(exact, want, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(fast, want, rtol=1e-12, atol=1e-280)     # (results below 1e-280 are exp's subnormal tail)
    assert np.array_equal(fast == 0.0, want == 0.0) or np.max(np.abs(fast - want)) < 1e-300


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "inf_*.npz"))), ids=os.path.basename)
def test_inner_protocol_debug_export_vs_reference_raw_dict(path):
    """hmogp_debug_raw_grads vs the gradient dict the reference's SVMOGPInf.inference returned for the same inputs
    (svmogp_inf.py:107,130-171) -- nothing of the oracle in between."""
    from oracle import svmogp_oracle as so
    g = np.load(path)
    prm, prob, X, Y, bs = so.load_case(g)          # (load_case only rebuilds the input arrays of the fixture)
    prm.pop("W0", None)
    e = make_engine(prob, X, Y)
    out = run(e, prm, bs)
    assert rel(out["elbo"], g["elbo"]) < 1e-8
    raw = e.debug_raw_grads([x.shape[0] for x in X])
    Q, Df = prob["Q"], prob["Df"]
    for q in range(Q):
        assert rel(raw["dL_dKmm"][q], g["dL_dKmm_%d" % q]) < 1e-8, ("dL_dKmm", q)
        for d in range(Df):
            assert raw["dL_dKmn"][q][d].shape == g["dL_dKmn_%d_%d" % (q, d)].shape
            assert rel(raw["dL_dKmn"][q][d], g["dL_dKmn_%d_%d" % (q, d)]) < 1e-8, ("dL_dKmn", q, d)
            assert rel(raw["dL_dKdiag"][q][d], np.ravel(g["dL_dKdiag_%d_%d" % (q, d)])) < 1e-8, ("dL_dKdiag", q, d)


def test_debug_export_refuses_partial_evaluations():
    from hetmogp_amd import _lib
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {})]
    prm, prob, X, Y = synth(31, specs, [300, 200], 16, 2, 1, (1.0, 1.2))
    e = make_engine(prob, X, Y)
    run(e, prm, group_mask=_lib.GROUP_QU)
    with pytest.raises(_lib.HetMOGPError):
        e.debug_raw_grads([300, 200])
    e2 = make_engine(prob, X, Y, chunk_rows=128)   # several pools: P~ of the first pools is gone
    run(e2, prm)
    with pytest.raises(_lib.HetMOGPError):
        e2.debug_raw_grads([300, 200])


def test_forced_ladder_at_M512_matches_oracle_at_equal_rung():
    """M = 512 with lengthscale = 4 inducing spacings: plain dpotrf fails robustly, GPy's ladder is TAKEN;
    CPU and GPU must land on the same rung, and are compared at that rung."""
    from oracle import svmogp_oracle as so
    specs = [("Gaussian", {"sigma": 0.5}), ("Poisson", {})]
    prm, prob, X, Y = synth(32, specs, [700, 600], 512, 2, 1, (4.0, 4.5))
    prm["Z"] = np.tile(np.linspace(0, 1, 512)[:, None], (1, 2))
    want = so.elbo_grad_fused(prm, prob, X, Y)
    assert min(want["rungs"]) >= 0
    e = make_engine(prob, X, Y)
    free = run(e, prm)
    assert free["rungs"] == want["rungs"]
    out = run(e, prm, forced_rung=want["rungs"])
    lit = so.elbo_grad_literal(prm, prob, X, Y, forced_rungs=want["rungs"])
    for k in KEYS:      # conditioning-limited: yardstick = distance between the oracle's own two restatements
        assert rel(out[k], want[k]) < max(1e-5, 10.0 * rel(want[k], lit[k])), k


def test_wire_format_roundtrip_and_sum():
    """The exchange step in its wire format: pack -> (sum of two shards' wires) -> unpack -> finish equals one pass."""
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Gamma", {})]
    Ns = [900, 700, 500]
    prm, prob, X, Y = synth(33, specs, Ns, 130, 2, 1, (0.9, 1.2))
    args = dict(Z=prm["Z"], m_u=prm["m_u"], L_flat=prm["L_flat"], variance=prm["variance"], lengthscale=prm["lengthscale"],
                W=prm["W"], kappa=prm["kappa"])
    e = make_engine(prob, X, Y)
    full = e.elbo_grad(**args)
    _, n = e.wire_buffer()
    Q, M, P, Df = prob["Q"], prob["M"], prob["P"], prob["Df"]
    assert n == 2 + Df + Q * (M * (M + 1) // 2 + M + M * P + 2 + Df)
    cut = [n_ // 2 for n_ in Ns]
    e.step_begin(row_begin=[0] * 3, row_end=cut, **args)
    e.wire_pack()
    w1 = e.wire_read()
    e.step_begin(row_begin=cut, row_end=Ns, **args)
    e.wire_pack()
    w2 = e.wire_read()
    e.wire_write(w1 + w2)
    e.wire_unpack()
    both = e.step_finish()
    for k in KEYS:
        assert rel(both[k], full[k]) < 1e-9, k
    # pack -> unpack is the identity on what finish reads
    e.step_begin(**args)
    e.wire_pack()
    e.wire_unpack()
    again = e.step_finish()
    for k in KEYS:
        assert np.array_equal(np.asarray(again[k]), np.asarray(full[k])), k


# ------------------------------------------------------------------------------------------------ quirks = "exact"
EXACT_CASES = [
    ("cat_gauss", [("Categorical", {"K": 3}), ("Gaussian", {"sigma": 0.5})], [160, 200], 20, 2, 1),
    ("gamma_beta", [("Gamma", {}), ("Beta", {}), ("Bernoulli", {})], [150, 140, 130], 24, 2, 1),
    ("eight_2d", [("HetGaussian", {}), ("Categorical", {"K": 4}), ("Poisson", {}), ("Exponential", {})], [90, 80, 70, 60], 25, 3, 2),
]


@pytest.mark.parametrize("tag,specs,Ns,M,Q,P", EXACT_CASES, ids=[c[0] for c in EXACT_CASES])
def test_exact_mode_matches_oracle_and_finite_differences(tag, specs, Ns, M, Q, P):
    """quirks = 0: (a) equals the oracle's exact mode; (b) every returned gradient is the gradient of the returned ELBO
    (central differences through the C ABI, every parameter group incl. Categorical m_u and W), tolerance 1e-5."""
    from oracle import svmogp_oracle as so
    prm, prob, X, Y = synth(40 + len(tag), specs, Ns, M, Q, P, (1.0, 1.2, 0.9)[:Q])
    prm["kappa"] = 0.05 + 0.02 * np.arange(prob["Q"] * prob["Df"], dtype=float).reshape(prob["Q"], prob["Df"])
    prm["W0"] = prm["W"] * 3.0                     # must be IGNORED in exact mode (quirk Q3)
    want = so.elbo_grad_fused({k: v for k, v in prm.items() if k != "W0"}, dict(prob, quirks="exact"), X, Y)
    e = make_engine(prob, X, Y, quirks="exact")
    out = run(e, prm)
    for k in KEYS:
        assert rel(out[k], want[k]) < 1e-8, k
    ref_mode = run(make_engine(prob, X, Y), prm)   # and the reference mode differs where the quirks bite
    assert rel(ref_mode["g_W"], out["g_W"]) > 1e-3
    rng = np.random.RandomState(5)
    for key, gkey in (("m_u", "g_m_u"), ("L_flat", "g_L_u"), ("lengthscale", "g_lengthscale"), ("Z", "g_Z"),
                      ("variance", "g_variance"), ("W", "g_W"), ("kappa", "g_kappa")):
        d = rng.randn(*np.shape(prm[key]))
        # (L_flat: the ELBO's rounding noise, ~1e-14 relative through an ill-conditioned K_uu, divided by 2 eps is 2e-6 .. 1e-5
        #  of this derivative at the 1e-6 step on every path -- tools/fd_noise.py; the 1e-5 step has 10x less of it and a
        #  truncation error far below the tolerance)
        eps = {"Z": 1e-7, "L_flat": 1e-5}.get(key, 1e-6) * np.abs(prm[key]).max()
        p1, p2 = dict(prm), dict(prm)
        p1[key], p2[key] = prm[key] + eps * d, prm[key] - eps * d
        fd = (run(e, p1)["elbo"] - run(e, p2)["elbo"]) / (2 * eps)
        an = float(np.sum(np.asarray(out[gkey]) * d))
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)) + 2e-4 * abs(an) * (key == "Z"), (key, fd, an)


@pytest.mark.parametrize("name,kw", [("Gamma", {}), ("Beta", {}), ("Categorical", {"K": 3}), ("Categorical", {"K": 5})])
def test_var_exp_exact_mode_is_differentiable(name, kw):
    """hmogp_var_exp_ex with quirks = 0: dm and dv are the derivatives of ve (finite differences in m and v)."""
    from hetmogp_amd import engine as E
    rng = np.random.RandomState(3)
    J = E.lik_dim_f(name, **kw)
    N = 40
    m, v = 0.6 * rng.randn(N, J), 0.2 + 0.5 * rng.rand(N, J)
    if name == "Gamma":
        y = rng.gamma(2.0, 1.0, N) + 1e-2
    elif name == "Beta":
        y = np.clip(rng.beta(2.0, 3.0, N), 1e-3, 1 - 1e-3)
    else:
        y = rng.randint(1, kw["K"] + 1, N).astype(float)
    ve, dm, dv = E.var_exp(name, y, m, v, quirks="exact", **kw)
    h = 1e-6
    for j in range(J):
        e = np.zeros((1, J))
        e[0, j] = h
        fm = (E.var_exp(name, y, m + e, v, quirks="exact", **kw)[0] - E.var_exp(name, y, m - e, v, quirks="exact", **kw)[0]) / (2 * h)
        fv = (E.var_exp(name, y, m, v + e, quirks="exact", **kw)[0] - E.var_exp(name, y, m, v - e, quirks="exact", **kw)[0]) / (2 * h)
        assert np.max(np.abs(fm - dm[:, j])) < 1e-6 * max(1.0, np.max(np.abs(dm[:, j])))
        # dv = 1/2 E[d2 log p / df2] equals d ve / d v exactly for Gaussian q(f) (Price's theorem); the GH rule integrates
        # both sides approximately, so this one holds to quadrature accuracy only
        assert np.max(np.abs(fv - dv[:, j])) < 2e-2 * max(1.0, np.max(np.abs(dv[:, j])))
    ref = E.var_exp(name, y, m, v, **kw)
    if name in ("Gamma", "Beta"):
        assert rel(np.pi * ref[0], ve) < 1e-13 and rel(np.pi * ref[1], dm) < 1e-13     # quirk Q1 is exactly a factor 1/pi
    else:
        assert rel(ref[0], ve) < 1e-14 and rel(ref[2], dv) < 1e-14                       # quirk Q2 only touches dm


def test_vem_converges_higher_with_exact_gradients():
    """L-BFGS VEM on a [Categorical(3), Gaussian] toy: with the reference's Categorical d/dm (quirk Q2) and W gradient
    (Q4) the line searches see inconsistent gradients; with quirks = "exact" the same driver reaches a higher ELBO."""
    import hetmogp_amd as H
    rng = np.random.RandomState(7)
    N, M, Q = 300, 12, 2
    X = [np.sort(rng.rand(N, 1), 0), np.sort(rng.rand(N, 1), 0)]
    f = lambda x: np.hstack([2.0 * np.sin(6 * x), 2.0 * np.cos(5 * x)])
    F0 = f(X[0])
    e = np.exp(F0)
    p = np.hstack([e, np.ones((N, 1))]) / (1 + e.sum(1, keepdims=True))
    Y = [(1 + (rng.rand(N, 1) > np.cumsum(p, 1)).sum(1, keepdims=True)).clip(1, 3).astype(float),
         np.sin(8 * X[1]) + 0.3 * rng.randn(N, 1)]
    elbo = {}
    for mode in ("reference", "exact"):
        np.random.seed(11)
        lik = H.HetLikelihood([H.Categorical(3), H.Gaussian(sigma=0.3)])
        md = lik.generate_metadata()
        kern = H.latent_functions_prior(Q, lenghtscale=[0.15, 0.2], variance=[1.0, 1.0], input_dim=1)
        model = H.SVMOGP(X=X, Y=Y, Z=np.linspace(0, 1, M)[:, None], kern_list=kern, likelihood=lik, Y_metadata=md,
                         quirks=mode)
        model.q_u_means[...] = 0.1 * np.random.randn(M, Q)
        H.vem_algorithm(model, stochastic=False, vem_iters=3)
        if mode == "exact":
            elbo[mode] = float(model.log_likelihood()[0, 0])
        else:      # compare like with like: the ELBO of the reference-mode optimum (Q2 does not change the ELBO itself)
            elbo[mode] = float(model.log_likelihood()[0, 0])
    assert np.isfinite(elbo["exact"]) and elbo["exact"] > elbo["reference"]


# ------------------------------------------------------------------------------------------------ f4: device samples
SAMPLE_CASES = [("Gaussian", {"sigma": 0.7}), ("Bernoulli", {}), ("HetGaussian", {}), ("Poisson", {}), ("Exponential", {}),
                ("Gamma", {}), ("Beta", {}), ("Categorical", {"K": 4})]


@pytest.mark.parametrize("name,kw", SAMPLE_CASES, ids=[c[0] for c in SAMPLE_CASES])
def test_device_samples_have_the_reference_distribution(name, kw):
    """hmogp_sample (the reference's `<likelihood>.samples`, e.g. gamma.py:43-50, categorical.py:65-75): the generator is a
    different stream than NumPy's, so the test is statistical -- per distinct f, 40 000 draws must reproduce the mean and
    variance the reference's link functions imply (5 sigma of the Monte-Carlo error), plus the supports."""
    from hetmogp_amd import engine as E
    J = E.lik_dim_f(name, **kw)
    S = 40000
    rng = np.random.RandomState(11)
    fs = [rng.uniform(-1.5, 1.5, J) for _ in range(4)]
    if name == "Poisson":
        fs += [np.array([3.5]), np.array([6.0])]                      # lambda = 33 and 403: the PTRS branch
    for f in fs:
        F = np.tile(f[None, :], (S, 1))
        y = E.sample(name, F, seed=int(rng.randint(1 << 30)), **kw)
        assert y.shape == (S, 1) and np.all(np.isfinite(y))
        y = y[:, 0]
        ef = np.exp(f)
        if name == "Gaussian":
            mean, var = f[0], kw["sigma"] ** 2
        elif name == "Bernoulli":
            p = ef[0] / (1 + ef[0])
            mean, var = p, p * (1 - p)
            assert set(np.unique(y)) <= {0.0, 1.0}
        elif name == "HetGaussian":
            mean, var = f[0], ef[1]
        elif name == "Poisson":
            mean, var = ef[0], ef[0]
            assert np.all(y >= 0) and np.all(y == np.floor(y))
        elif name == "Exponential":
            mean, var = np.exp(-f[0]), np.exp(-2 * f[0])              # scale = exp(-f) (exponential.py:52-56)
            assert np.all(y > 0)
        elif name == "Gamma":
            mean, var = ef[0] / ef[1], ef[0] / ef[1] ** 2             # shape a = e^f1, rate b = e^f2 (gamma.py:43-50)
            assert np.all(y > 0)
        elif name == "Beta":
            a, b = ef
            mean, var = a / (a + b), a * b / ((a + b) ** 2 * (a + b + 1))
            assert np.all((y >= 0) & (y <= 1))                      # (x/(x+y) rounds to the end points for tiny shapes)
        else:
            K = kw["K"]
            p = np.append(ef, 1.0) / (1 + ef.sum())
            assert set(np.unique(y)) <= set(float(k) for k in range(1, K + 1))      # labels 1..K (categorical.py:81)
            freq = np.array([(y == k + 1).mean() for k in range(K)])
            assert np.all(np.abs(freq - p) < 5 * np.sqrt(p * (1 - p) / S) + 1e-4)
            continue
        m4 = max(3.0 * var ** 2, 1e-12)                                 # crude bound on the 4th central moment
        assert abs(y.mean() - mean) < 5 * np.sqrt(var / S) + 1e-12, (name, f, y.mean(), mean)
        assert abs(y.var() - var) < 8 * np.sqrt(m4 / S) * (4 if name in ("Exponential", "Gamma", "Poisson") else 1), (name, f)


def test_device_samples_are_reproducible_per_seed():
    from hetmogp_amd import engine as E
    F = np.linspace(-1, 1, 64)[:, None]
    a, b, c = E.sample("Poisson", F, seed=5), E.sample("Poisson", F, seed=5), E.sample("Poisson", F, seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_het_likelihood_samples_run_on_the_device():
    import hetmogp_amd as H
    lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Categorical(3), H.Gamma()])
    md = lik.generate_metadata()
    rng = np.random.RandomState(0)
    F = [rng.randn(50, 1), rng.randn(50, 2), 0.3 * rng.randn(50, 2)]
    np.random.seed(4)
    Y1 = lik.samples(F, md)
    np.random.seed(4)
    Y2 = lik.samples(F, md)
    assert [y.shape for y in Y1] == [(50, 1)] * 3 and all(np.array_equal(a, b) for a, b in zip(Y1, Y2))
    assert set(np.unique(Y1[1])) <= {1.0, 2.0, 3.0} and np.all(Y1[2] > 0)


# ------------------------------------------------------------------------------------------------ specialised row-pass GEMMs
@pytest.mark.parametrize("M,Ns,P", [(128, [777, 130, 1], 1), (256, [1030, 515], 2), (320, [900, 333], 1), (384, [2049, 17, 640], 1),
                                    (512, [1500, 700], 1)])     # (512: four column tiles = two fold pairs, diagonal-block skipping)
def test_specialised_rowpass_kernels_vs_oracle(M, Ns, P):
    """gemm_rowpass.hip (8-wave forward / Gram kernels, taken when the inducing dimension is a multiple of 128) with ragged
    row counts (not multiples of 128 nor of 16, a 1-row task), full gradients and the E-step's triangular fold, against
    the oracle; and several pools (row chunks) against one.  M >= 256 and a multiple of 64 also routes the replicated
    M x M products through gemm_small.hip (M = 320: general row-pass kernels + 64 x 64-tile M x M products)."""
    from oracle import svmogp_oracle as so
    from hetmogp_amd import _lib
    specs = [("Gaussian", {"sigma": 0.5}), ("Poisson", {}), ("Bernoulli", {})][:len(Ns)]
    Q = 2
    prm, prob, X, Y = synth(50 + M, specs, Ns, M, Q, P, (1.0, 1.25))
    want = so.elbo_grad_fused(prm, prob, X, Y)
    e = make_engine(prob, X, Y)
    out = run(e, prm)
    for k in KEYS:
        assert rel(out[k], want[k]) < 1e-8, k
    qu = run(e, prm, group_mask=_lib.GROUP_QU)            # forward against tril-folded C (b_tri), no P~ store
    for k in ("elbo", "g_m_u", "g_L_u"):
        assert rel(qu[k], want[k]) < 1e-8, k
    hz = run(e, prm, group_mask=_lib.GROUP_HYPER)         # hyper statistics without the Z gradient: P~ never stored
    for k in ("elbo", "g_variance", "g_lengthscale", "g_W"):
        assert rel(hz[k], want[k]) < 1e-8, k
    e2 = make_engine(prob, X, Y, chunk_rows=300)
    chunked = run(e2, prm)
    for k in KEYS:          # summation order only; the 2-D case has cond(K_uu) ~ 1e6 (random inducing jitter), hence 5e-8
        assert rel(chunked[k], out[k]) < 5e-8 and rel(chunked[k], want[k]) < 5e-8, k
    m, v = e.predict_f(X[0])
    u = so.u_algebra(prm, prob)
    for d in range(prob["Df"]):
        if prob["f_index"][d] != 0:
            continue
        mm_, vv_ = 0.0, 0.0
        for q in range(Q):
            K = so.rbf_K(X[0], prm["Z"][:, q * P:(q + 1) * P], prm["variance"][q], prm["lengthscale"][q])
            w = prm["W"][q, d]
            mm_ = mm_ + w * (K @ u["a"][q])
            vv_ = vv_ + (w * w + prm["kappa"][q, d]) * prm["variance"][q] + w * w * np.sum((K @ u["C"][q]) * K, 1)
        assert rel(m[:, d], mm_) < 1e-8 and rel(v[:, d], vv_) < 2e-7    # (v = prior - explained: cancellation x cond(K_uu))
