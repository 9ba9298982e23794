import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("FUZZ_PKG_ROOT"):          # an older build of the package (A/B runs): <dir>/hetmogp_amd
    sys.path.insert(0, os.environ["FUZZ_PKG_ROOT"])
import numpy as np
import test_gpu_fuzz as fz
from test_gpu_engine import KEYS, make_engine, rel, run, synth
from oracle import svmogp_oracle as so
seed = int(sys.argv[1])
rng = np.random.RandomState(1000 + seed)
ms = fz.MS
M = ms[seed % len(ms)]
P = 1 if (rng.rand() < 0.7) else 2
Q = int(rng.randint(1, 4)); T = int(rng.randint(1, 4))
specs = [fz.LIKS[i] for i in rng.choice(len(fz.LIKS), T, replace=False)]
Ns = [int(rng.choice([0, 1, 17, 130, 257, 700, 1500], p=[0.05, 0.05, 0.1, 0.2, 0.2, 0.2, 0.2])) for _ in range(T)]
if sum(Ns) == 0: Ns[0] = 129
cs = tuple(0.9 + 0.4 * rng.rand(Q))
prm, prob, X, Y = synth(2000 + seed, specs, Ns, M, Q, P, cs)
quirks = "exact" if seed % 3 == 0 else "reference"
prob = dict(prob, quirks=quirks)
want = so.elbo_grad_fused(prm, prob, X, Y)
chunk = int(rng.choice([1 << 20, 600, 256]))
e = make_engine(prob, X, Y, chunk_rows=chunk, quirks=quirks)
out = run(e, prm)
print("seed", seed, "M", M, "P", P, "Q", Q, specs, Ns, "chunk", chunk, "quirks", quirks, "cond", out["cond_est"], "rungs", out["rungs"])
print({k: "%.2e" % rel(out[k], want[k]) for k in KEYS})
lit = so.elbo_grad_literal(prm, prob, X, Y) if quirks == "reference" and min(Ns) > 0 else None
if lit is not None:
    print("fused vs literal:", {k: "%.2e" % rel(want[k], lit[k]) for k in KEYS})
e2 = make_engine(prob, X, Y, quirks=quirks)
out2 = run(e2, prm)
print("one pool:", {k: "%.2e" % rel(out2[k], want[k]) for k in KEYS})
d = np.abs(np.asarray(out["g_Z"]) - np.asarray(want["g_Z"]))
idx = np.argsort(d.ravel())[::-1][:6]
print("max |g_Z|", np.max(np.abs(want["g_Z"])))
for i in idx:
    m, c = np.unravel_index(i, d.shape)
    print("  Z[%d,%d]=%.17g  engine %.12e oracle %.12e diff %.3e" % (m, c, prm["Z"][m, c], np.asarray(out["g_Z"])[m, c], np.asarray(want["g_Z"])[m, c], d[m, c]))
# is an inducing point (numerically) on top of a data point?
for t, x in enumerate(X):
    if len(x):
        dd = np.abs(x[:, 0][:, None] - prm["Z"][:, 0][None, :])
        j = np.unravel_index(np.argmin(dd), dd.shape)
        print("  task %d: min |x - z| = %.3e (row %d, inducing %d)" % (t, dd[j], j[0], j[1]))
zz = np.abs(prm["Z"][:, 0][:, None] - prm["Z"][:, 0][None, :]) + np.eye(M) * 1e9
print("  min |z - z'| = %.3e" % zz.min())
