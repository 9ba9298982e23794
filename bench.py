#!/usr/bin/env python
"""bench.py -- ELBO-steps/sec of the svmogp_inf hot path on MI355X (BASELINE.json metric).

One step = one full `parameters_changed()` equivalent: ELBO + every parameter gradient (m_u, L_u, RBF variance /
lengthscale, W, kappa, Z), inputs resident in HBM, parameters re-uploaded and gradients copied back every step.
Workload (config.workload): the headline config H of BASELINE.md -- T=4 [Gaussian, Bernoulli, Poisson, Gamma]
(Df=5), N_t=200 000 rows per task, M=1024 inducing points, Q=3 latent GPs, P=1, synthetic data.

  python bench.py --gpus N --steps K --warmup W
N>1 is launched by torch.distributed.run (one rank per GPU, RCCL): the rows of every task are sharded over the ranks
(strong scaling: total work fixed), each rank runs `hmogp_step_begin` on its rows, the statistic bundle is
sum-all-reduced once per step, `hmogp_step_finish` runs replicated.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      FP64-MFMA roofline of the dominant kernel (forward N x M x M contraction P~ = K^ C_q)
  roofline_kuf  HBM roofline of K_uf construction (rbf_cross_cov), the kernel the north-star singles out
  cpu_baseline  the NumPy/BLAS oracle ("port") timed on this host's cores on a bounded row sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPECS = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet FP64 matrix peak (the microarch guide lists no FP64 row)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)


class _StdoutToStderr(object):
    """RCCL prints its version banner to stdout when the first communicator is created; the contract is ONE JSON line on
    stdout, so everything before the result (process-group set-up, warm-up) runs with fd 1 pointed at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=200000, help="rows per task (headline: 200000)")
    ap.add_argument("--inducing", type=int, default=1024, help="M (headline: 1024)")
    ap.add_argument("--latents", type=int, default=3, help="Q (headline: 3)")
    ap.add_argument("--cpu-sample-rows", type=int, default=4000, help="rows per task of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-zero-pass", action="store_true", help="skip the extra (untimed-for-value) opt-in mode pass")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    guard = _StdoutToStderr()
    guard.__enter__()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from hetmogp_amd.engine import Engine
    from hetmogp_amd.synthetic import make_case
    from hetmogp_amd import dist as hdist

    N, M, Q, P, T = args.rows, args.inducing, args.latents, 1, len(SPECS)
    prm, X, Y = make_case(SPECS, [N] * T, M=M, Q=Q, P=P, seed=20260929)
    eng = Engine(SPECS, Q, M, P, device=local_rank, reuse_outputs=True)   # gradients land in page-locked arrays
    eng.set_data(X, Y)                      # every rank holds the (tiny) raw data; it only touches its own rows
    from hetmogp_amd.engine import pinned_empty
    for k in ("Z", "m_u", "L_flat"):        # the optimiser's parameter vectors live in page-locked host memory: H2D by DMA
        a = pinned_empty(np.shape(prm[k]))
        a[...] = prm[k]
        prm[k] = a
    reducer = hdist.StatsReducer(eng, device=local_rank) if world > 1 else None
    rb, re = hdist.shard_ranges([0] * T, [N] * T, rank, world)

    def step():
        if world == 1:
            return eng.elbo_grad(**prm)
        eng.step_begin(row_begin=rb, row_end=re, **prm)
        reducer()
        return eng.step_finish()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    guard.__exit__()
    t0 = time.perf_counter()
    cat_ms, cat_n = {}, {}
    for _ in range(args.steps):
        out = step()
        ms, nl = eng.timings()              # HIP-event spans on the engine's stream, per kernel family
        for k in ms:
            cat_ms[k] = cat_ms.get(k, 0.0) + ms[k]
            cat_n[k] = cat_n.get(k, 0) + nl[k]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if not np.isfinite(out["elbo"]):
        raise SystemExit("bench.py: non-finite ELBO")

    if rank == 0:
        rows_rank = sum(e - b for b, e in zip(rb, re))          # rows this rank streamed per step (all tasks)
        pairs_rows = rows_rank * Q                               # (row, latent) pairs per step
        # dominant kernel: forward contraction; one launch = all Q latents of one task chunk = 2*n*M*M*Q algorithmic
        # flops; the category holds exactly that kernel, so cat_ms / launches = its average duration
        fwd_flops = 2.0 * pairs_rows * M * M * args.steps
        fwd_s = cat_ms["forward_gemm"] / 1e3
        achieved = fwd_flops / fwd_s / 1e12 if fwd_s > 0 else 0.0
        kuf_bytes = 8.0 * pairs_rows * M * args.steps            # K_uf: N*M*8 bytes written per (task, latent)
        kuf_s = cat_ms["rbf_cross_cov"] / 1e3
        kuf_gbs = kuf_bytes / kuf_s / 1e9 if kuf_s > 0 else 0.0
        gram_flops = 1.0 * pairs_rows * M * M * args.steps       # lower-triangular weighted Gram: n*M*M
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_forward_gemm.json")
        if os.path.exists(pmc) and (N, M, Q) == (200000, 1024, 3) and world == 1:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        line = {
            "metric": "ELBO-steps/sec (one step = ELBO + all parameter gradients)",
            "value": args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "H: T=4 [Gaussian,Bernoulli,Poisson,Gamma] Df=5, N_t=%d rows/task, M=%d, Q=%d, P=1, "
                                   "full-batch ELBO+gradients" % (N, M, Q),
                       "rows_per_task": N, "M": M, "Q": Q, "T": T, "sharding": "rows/%d" % world},
            "roofline": {"kernel": "gemm_f64_kernel<false, true, 1> (forward P~ = K^ C_q + fused row statistics)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "launches": cat_n["forward_gemm"], "avg_launch_ms": cat_ms["forward_gemm"] / max(cat_n["forward_gemm"], 1)},
            "roofline_kuf": {"kernel": "rbf_kernel<1, false> (K_uf construction)", "bound": "hbm", "achieved": kuf_gbs,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": kuf_gbs / PEAK_HBM_GBS, "traffic": None,
                             "launches": cat_n["rbf_cross_cov"],
                             "avg_launch_ms": cat_ms["rbf_cross_cov"] / max(cat_n["rbf_cross_cov"], 1)},
            "kernel_ms_per_step": {k: v / args.steps for k, v in cat_ms.items()},
            # the weighted Gram shares the device with the HBM-bound column statistics (second stream): its span is longer
            # than when it runs alone, and gram_gemm + colstats_reduce overlap (they do not add up to the wall time)
            "gram_tflops": gram_flops / (cat_ms["gram_gemm"] / 1e3) / 1e12 if cat_ms["gram_gemm"] > 0 else 0.0,
            "elbo": out["elbo"],
        }
        if world == 1 and not args.no_exact_zero_pass:
            line["exact_zero_windows"] = exact_zero_pass(args, prm, X, Y, N, M, Q, P, out)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, prm, X, Y, N, M, Q, P)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def exact_zero_pass(args, prm, X, Y, N, M, Q, P, dense_out):
    """Extra information, NOT the reported `value`: the same workload with HMOGP_CFG_EXACT_ZERO_WINDOWS (opt-in).  The
    sorted 1-D inputs make K_uf banded (exp underflows to exactly 0.0 beyond ~38.6 lengthscales); the engine then skips
    products with exact zeros.  Results are compared with the dense pass of this run."""
    import numpy as np
    import torch
    from hetmogp_amd.engine import Engine
    eng = Engine(SPECS, Q, M, P, device=0, exact_zero_windows=True)
    eng.set_data(X, Y)
    for _ in range(max(1, args.warmup)):
        out = eng.elbo_grad(**prm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cat = {}
    for _ in range(args.steps):
        out = eng.elbo_grad(**prm)
        ms, _ = eng.timings()
        for k in ms:
            cat[k] = cat.get(k, 0.0) + ms[k]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    worst = 0.0
    for k in ("elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"):
        a, b = np.asarray(out[k], float), np.asarray(dense_out[k], float)
        worst = max(worst, float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300)))
    return {"value": args.steps / el, "unit": "steps/s", "ms_per_step": 1e3 * el / args.steps,
            "max_rel_diff_vs_dense": worst, "kernel_ms_per_step": {k: v / args.steps for k, v in cat.items()},
            "note": "opt-in mode; bit-for-bit the same terms minus products with exact 0.0; not used for `value`"}


def cpu_baseline(args, prm, X, Y, N, M, Q, P):
    """The oracle's fused restatement (same algebra and flop count as the GPU path: NumPy + multithreaded BLAS, fp64) on
    the first `cpu_sample_rows` rows of every task; steps/s is scaled linearly to the full row count (the M^3 part,
    which does not shrink with the sample, is charged in full -- conservative in the CPU's favour for value)."""
    import numpy as np
    from oracle import svmogp_oracle as so
    ns = min(args.cpu_sample_rows, N)
    prob = so.make_problem(SPECS, Q, M, P)
    Xs, Ys = [x[:ns] for x in X], [y[:ns] for y in Y]
    so.elbo_grad_fused(prm, prob, Xs, Ys)                       # warm-up (BLAS thread pool, page faults)
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        so.elbo_grad_fused(prm, prob, Xs, Ys)
        times.append(time.perf_counter() - t0)
    t = min(times)
    return {"value": 1.0 / (t * (float(N) / ns)), "unit": "steps/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "oracle.svmogp_oracle.elbo_grad_fused (NumPy+BLAS fp64, all host cores) on the first %d of %d rows of "
                      "each of the 4 tasks, M=%d, Q=%d: %.2f s per sampled step, scaled by %d/%d to the full step"
                      % (ns, N, M, Q, t, N, ns),
            "sample_seconds_per_step": t}


if __name__ == "__main__":
    main()
