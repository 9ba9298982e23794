#!/usr/bin/env python
"""bench.py -- ELBO-steps/sec of the svmogp_inf hot path on MI355X (BASELINE.json metric).

One step = one full `parameters_changed()` equivalent: ELBO + every parameter gradient (m_u, L_u, RBF variance /
lengthscale, W, kappa, Z), inputs resident in HBM, parameters re-uploaded and gradients copied back every step.
Workload (config.workload): the headline config H of BASELINE.md -- T=4 [Gaussian, Bernoulli, Poisson, Gamma]
(Df=5), N_t=200 000 rows per task, M=1024 inducing points, Q=3 latent GPs, P=1, synthetic data.

  python bench.py --gpus N --steps K --warmup W
N>1 is launched by torch.distributed.run (one rank per GPU, RCCL): the rows of every task are sharded over the ranks
(strong scaling: total work fixed), each rank uploads and streams only its rows (`hmogp_step_begin`), the statistic
bundle is sum-all-reduced once per step in its wire format (lower triangles of H_q, 12.7 MB), `hmogp_step_finish` runs
replicated.  `python bench.py --gpus N` without a rendezvous in the environment launches itself under
torch.distributed.run.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      FP64-MFMA roofline of the dominant kernel (forward N x M x M contraction P~ = K^ C_q)
  roofline_kuf  HBM roofline of K_uf construction (rbf_cross_cov), the kernel the north-star singles out
  cpu_baseline  baseline B: the NumPy/BLAS oracle ("port", all host cores) timed at two row samples (N=1 only), next
                to the host's plain dgemm rate (host_dgemm_gflops) so that the port's efficiency is visible
  cpu_baseline_literal  baseline A: the literal reference algorithm (N x N terms) at C1 and up to N_t = 8192
  parity_at_headline_M  engine vs oracle on the sampled rows at the headline M (the bench fails above 1e-5)
  other_configs the other BASELINE.json configurations (C1, C2, C3 as the facade's SVI loop over N_all = 1M, the per-rank
                share of C4, C5 + predict_f on a 256 x 256 grid), a few steps each, with executed flops and the fraction
                of the FP64-MFMA peak (N=1 only; these are reported beside `value`, never mixed into it)
  N > 1: exchange_modes_ms_per_step times the step with the library's own RCCL communicator ("native", the default) and
         with the torch.distributed all-reduce on the aliased wire buffer ("device"); replicated_ms_per_rank.
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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _self_launch(ngpus):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: re-execute under torch.distributed.run,
    one rank per GPU (the driver may also launch the ranks itself, in which case RANK is set and this is skipped)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=200000, help="rows per task (headline: 200000)")
    ap.add_argument("--inducing", type=int, default=1024, help="M (headline: 1024)")
    ap.add_argument("--latents", type=int, default=3, help="Q (headline: 3)")
    ap.add_argument("--cpu-sample-rows", type=int, default=20000, help="rows per task of the larger CPU-baseline sample")
    ap.add_argument("--cpu-literal-budget", type=float, default=70.0, help="seconds the literal-reference baseline may use")
    ap.add_argument("--cpu-full-budget", type=float, default=240.0,
                    help="run ONE true full-size step of CPU baseline B if the two-sample prediction is below this many seconds")
    ap.add_argument("--no-other-configs", action="store_true", help="skip C1 / C2 / C3 / C4-share / C5 (N=1 only)")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps per other configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wakeup", action="store_true", help="skip the ~0.3 s synthetic device wake-up before the warm-up steps")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: run the DISTRIBUTED code path (nccl process group, StatsReducer negotiation, native RCCL "
                         "communicator, hmogp_elbo_grad_sharded, the alternative-exchange-mode loop, teardown order) with a world "
                         "of ONE rank -- a dry run of everything a multi-GPU launch executes, on a 1-GPU box")
    ap.add_argument("--no-exact-zero-pass", action="store_true", help="skip the extra (untimed-for-value) opt-in mode pass")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(_self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    guard = _StdoutToStderr()
    guard.__enter__()
    rccl_ranks = 1
    if args.force_dist and world != 1:
        raise SystemExit("bench.py: --force-dist is the one-rank dry run of the distributed path (use it with --gpus 1)")
    distm = world > 1 or args.force_dist       # the distributed code path (a world of one rank with --force-dist)
    if distm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones)                                   # proof that RCCL connected every rank
        rccl_ranks = int(round(float(ones.item())))
        if rccl_ranks != world:
            raise SystemExit("bench.py: RCCL all-reduce saw %d ranks, expected %d" % (rccl_ranks, world))

    from hetmogp_amd.engine import Engine
    from hetmogp_amd.synthetic import make_case
    from hetmogp_amd import dist as hdist

    N, M, Q, P, T = args.rows, args.inducing, args.latents, 1, len(SPECS)
    prm, X, Y = make_case(SPECS, [N] * T, M=M, Q=Q, P=P, seed=20260929)
    eng = Engine(SPECS, Q, M, P, device=local_rank, reuse_outputs=True)   # gradients land in page-locked arrays
    # strong scaling: the rows of every task are split into one contiguous range per rank; a rank uploads ONLY its rows
    rb, re = hdist.shard_ranges([0] * T, [N] * T, rank, world)
    eng.set_data([x[b:e] for x, b, e in zip(X, rb, re)], [y[b:e] for y, b, e in zip(Y, rb, re)])
    rows_rank = sum(e - b for b, e in zip(rb, re))              # rows this rank streams per step (all tasks)
    from hetmogp_amd.engine import pinned_empty
    for k in ("Z", "m_u", "L_flat"):        # the optimiser's parameter vectors live in page-locked host memory: H2D by DMA
        a = pinned_empty(np.shape(prm[k]))
        a[...] = prm[k]
        prm[k] = a
    if args.force_dist:
        hdist.force_collectives(True)      # the negotiation's flag reductions are issued for real, also with one rank
    reducer = hdist.StatsReducer(eng, device=local_rank, single_rank_exchange=args.force_dist) if distm else None

    # (the step closure, the timed loop, the exchange-mode sweep and the teardown live in hetmogp_amd/dist.py so that the CPU suite
    #  runs THIS control flow under gloo at world size 8 with a stand-in engine: tests/test_dist_cpu.py::test_world8_bench_control_flow)
    step = hdist.make_step(eng, prm, reducer, distm)

    def fence():
        if distm:
            dist.barrier()
        torch.cuda.synchronize()

    # Device wake-up (NOT a step of the workload): ~0.3 s of a synthetic FP64-MFMA contraction.  The first ~0.5 s of kernels after an
    # idle period run 5-10 % slower (clock ramp): without it a run with few warm-up steps times the ramp, not the path.
    if not args.no_wakeup:
        import ctypes as _C
        from hetmogp_amd._lib import lib as _hl
        _ms = _C.c_double()
        _hl.hmogp_bench_contraction(local_rank, 1, 131072, 1024, int(os.environ.get("HMOGP_WAKEUP_ITERS", "70")), _C.byref(_ms))
    # (the interpreter's cyclic garbage collector is kept out of the timed region: with torch imported a generation-2 collection
    #  walks ~1e6 objects and stalls the host thread for 30-40 ms -- seen as ONE 158 ms step among twenty 120.7 ms ones)
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        out = step()
    fence()
    guard.__exit__()
    elapsed, cat_ms, cat_n, step_walls, closing_fence_ms, out = hdist.timed_steps(step, eng, args.steps, fence, reducer)
    # (the collector stays off for the other timed loops of this process; they collect explicitly between workloads)
    rows_all = [rows_rank]
    fwd_rank = [[cat_ms.get("forward_gemm", 0.0), float(cat_n.get("forward_gemm", 0)), float(rows_rank)]]
    if distm:
        elapsed = hdist.max_over_ranks(elapsed, local_rank)
        fwd_rank = hdist.gather_floats(fwd_rank[0], world, local_rank)
        rows_all = [int(r[2]) for r in fwd_rank]
    if not np.isfinite(out["elbo"]):
        raise SystemExit("bench.py: non-finite ELBO")

    # ---- N > 1: the same steps with the other exchange modes (reported, not `value`) ------------------------------------
    exchange_modes, repl_all = {}, None
    if distm:
        exchange_modes = hdist.exchange_mode_sweep(
            eng, step, reducer, args.steps, args.warmup, fence, local_rank, elapsed, cat_ms,
            make_reducer=lambda alt: hdist.StatsReducer(eng, device=local_rank, mode=alt, single_rank_exchange=args.force_dist))
        repl_all = [r[0] for r in hdist.gather_floats([cat_ms["mxm_algebra"] / args.steps], world, local_rank)]

    if rank == 0:
        pairs_rows = rows_rank * Q                               # (row, latent) pairs per step on this rank
        # dominant kernel: forward contraction; one launch = all Q latents of all rows of the step (800 000 at the headline
        # size) = 2 * rows * Q * M * M algorithmic flops; the category holds exactly that kernel, so
        # cat_ms / launches = its average launch duration (HIP events on the engine's stream around every launch)
        fwd_flops = 2.0 * pairs_rows * M * M * args.steps
        fwd_s = cat_ms["forward_gemm"] / 1e3
        achieved = fwd_flops / fwd_s / 1e12 if fwd_s > 0 else 0.0
        kuf_bytes = 8.0 * pairs_rows * M * args.steps            # K_uf: N*M*8 bytes written per (task, latent)
        kuf_s = cat_ms["rbf_cross_cov"] / 1e3
        kuf_gbs = kuf_bytes / kuf_s / 1e9 if kuf_s > 0 else 0.0
        gram_flops = 1.0 * pairs_rows * M * M * args.steps       # lower-triangular weighted Gram: n*M*M
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_forward_gemm.json")
        if os.path.exists(pmc) and (N, M, Q) == (200000, 1024, 3) and world == 1:
            rec = json.load(open(pmc))
            traffic, traffic_src = rec.get("hbm_bytes_per_launch"), rec.get("source")
        exec_flops = 3.0 * rows_rank * Q * M * M                # executed by the two contractions (forward 2nM^2 + Gram nM^2)
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
            "roofline": {"kernel": "rowpass_gemm_kernel<1> (forward P~ = K^ C_q + fused row statistics)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_measured_in_run": False,   # PMC passes need rocprofv3 around the process: profiles/
                         "launches": cat_n["forward_gemm"], "avg_launch_ms": cat_ms["forward_gemm"] / max(cat_n["forward_gemm"], 1)},
            # K_uf construction (the kernel the north-star singles out).  In the step it runs on the low-priority stream in
            # launches of 16384 rows interleaved with the latency-bound K_uu chain, so its in-step SPAN (in_step_*) includes
            # that chain; the roofline is quoted on the same kernel launched alone, one (task, all latents) launch shape
            # (kuf_alone(), HIP events around `iters` back-to-back launches).
            "roofline_kuf": kuf_alone(N, M, Q, kuf_gbs, cat_ms["rbf_cross_cov"] / args.steps, cat_n["rbf_cross_cov"] // args.steps),
            # the whole step against the FP64-MFMA peak on EXECUTED contraction flops (3 n M^2 per (row, latent) pair: the
            # Gram only forms lower tiles), this rank's share
            "step_tflops_executed": exec_flops / (elapsed / args.steps) / 1e12,
            "step_frac_of_peak_executed": exec_flops / (elapsed / args.steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "kernel_ms_per_step": {k: v / args.steps for k, v in cat_ms.items()},
            # the weighted Gram shares the device with the HBM-bound column statistics (second stream): its span is longer
            # than when it runs alone, and gram_gemm + colstats_reduce overlap (they do not add up to the wall time)
            "gram_tflops": gram_flops / (cat_ms["gram_gemm"] / 1e3) / 1e12 if cat_ms["gram_gemm"] > 0 else 0.0,
            "replicated_ms": cat_ms["mxm_algebra"] / args.steps,  # M x M algebra every rank repeats (the Amdahl term)
            "rccl_ranks": rccl_ranks,
            "step_wall_ms": [round(v, 3) for v in step_walls], "closing_fence_ms": round(closing_fence_ms, 3),
            "rows_per_rank": rows_all,
            "elbo": out["elbo"],
        }
        if reducer is not None:
            # "native" = ncclAllReduce issued by the library on the engine's stream (device time, HIP events around pack +
            # all-reduce + unpack); "device" = torch.distributed on the aliased wire buffer (host wall time incl. its syncs)
            line["allreduce_ms_per_step"] = exchange_modes[reducer.mode]["exchange_ms_per_step"]
            line["allreduce_frac_of_step"] = line["allreduce_ms_per_step"] / line["ms_per_step"]
            # the dominant kernel's roofline on EVERY rank (its own rows, its own HIP-event launch durations): a straggler shows here
            line["roofline_per_rank"] = [
                {"rank": r, "rows": int(rows_r), "launches": int(n_r), "avg_launch_ms": ms_r / max(n_r, 1.0),
                 "achieved": (2.0 * rows_r * Q * M * M * args.steps / (ms_r / 1e3) / 1e12) if ms_r > 0 else 0.0,
                 "frac": (2.0 * rows_r * Q * M * M * args.steps / (ms_r / 1e3) / 1e12 / PEAK_FP64_MFMA_TFLOPS) if ms_r > 0 else 0.0,
                 "unit": "TFLOP/s", "bound": "mfma", "peak": PEAK_FP64_MFMA_TFLOPS}
                for r, (ms_r, n_r, rows_r) in enumerate(fwd_rank)]
            line["replicated_frac_of_step"] = line["replicated_ms"] / line["ms_per_step"]
            line["n_gt_1_rccl_executed_before_this_run"] = False   # README: no N > 1 RCCL step had run anywhere before the driver's
            line["allreduce_bytes"] = 8 * int(eng.wire_buffer()[1])
            line["reducer_mode"] = reducer.mode
            line["exchange_modes_ms_per_step"] = exchange_modes
            line["replicated_ms_per_rank"] = repl_all
        if args.force_dist:
            line["force_dist"] = True       # one-rank dry run of the distributed path: no other configurations, no CPU legs
        if world == 1 and not distm and not args.no_exact_zero_pass:
            line["exact_zero_windows"] = exact_zero_pass(args, prm, X, Y, N, M, Q, P, out)
        # GPU legs first, CPU legs last: after the full-size CPU baseline (64 worker threads x BLAS threads, ~20 s of all host
        # cores) the launch-latency-bound chains of the small configurations measured 0.3 ms slower (host-side launch jitter)
        if world == 1 and not distm and not args.no_other_configs:
            line["other_configs"] = other_configs(args)   # (the headline engine keeps its 40 GB of row workspaces: 288 GB of HBM)
        if world == 1 and not distm and not args.no_cpu_baseline:
            line.update(cpu_baselines(args, eng, prm, X, Y, N, M, Q, P))
            eng.close()
        if "other_configs" in line:     # compact recap as the LAST key: a tail of the line still shows every configuration
            line["other_configs_summary"] = {c["workload"].split(":")[0].split(" (")[0]: [round(c["ms_per_step"], 4),
                                                                                         round(c.get("frac_of_peak", 0.0), 4)]
                                             for c in line["other_configs"]}
        print(json.dumps(line))
        sys.stdout.flush()
    if distm:
        hdist.teardown(reducer)             # ncclCommDestroy of the library's own communicator, on every rank, before torch's


def kuf_alone(N, M, Q, in_step_gbs, in_step_ms, in_step_launches):
    import ctypes as C
    from hetmogp_amd._lib import lib, check
    ms = C.c_double()
    check(lib.hmogp_bench_contraction(0, 5, int(N), int(M), 10, C.byref(ms)))
    gbs = 8.0 * N * M * 3 / (ms.value / 1e3) / 1e9              # role 5 batches 3 latents per launch, like the step's Q = 3
    return {"kernel": "rbf_kernel<1, false> (K_uf construction, one task x 3 latents per launch, alone)", "bound": "hbm",
            "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
            "avg_launch_ms": ms.value, "bytes_per_launch": 8.0 * N * M * 3,
            "in_step_span_ms": in_step_ms, "in_step_launches": in_step_launches, "in_step_span_GBps": in_step_gbs}


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


def _blas_info():
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            return info[0].get("internal_api", "?") + " " + str(info[0].get("version", "")), int(info[0].get("num_threads", 0))
    except Exception:
        pass
    return "unknown", 0


def _median_time(fn, reps):
    import statistics
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


def cpu_baselines(args, eng, prm, X, Y, N, M, Q, P):
    """CPU legs, on this host's cores, same inputs (oracle/ is the checker and the baseline, never the product):

    cpu_baseline          baseline B of BASELINE.md 4 ("port"): the oracle's fused restatement -- the same algebra and flop
        count as the GPU path, NumPy + multithreaded BLAS -- timed at TWO row samples (median of 5 steps each).  The cost
        splits into a part that does not depend on the rows (M x M algebra: fixed_s) and a part linear in them
        (per_row_s); only the latter is extrapolated to the full row count.
    parity_at_headline_M  max relative error (ELBO and the 7 gradient arrays) between the oracle on the larger sample and
        the engine on the same rows, at the headline M / Q / likelihood mix; the bench fails above 1e-5.
    cpu_baseline_literal  baseline A ("the reference CPU path"): the literal restatement with the N x N terms
        (oracle.inference_literal(full_cov=True) + assemble_literal) at C1 in full and at the largest N_t <= 8192 of the
        headline mix that fits the time budget."""
    import numpy as np
    from oracle import svmogp_oracle as so
    T = len(SPECS)
    blas, threads = _blas_info()
    prob = so.make_problem(SPECS, Q, M, P)
    ns2 = min(args.cpu_sample_rows, N)
    ns1 = max(1, ns2 // 4)
    reps_b = 3

    # calibration: what this host's BLAS does on a plain dgemm with all threads (the yardstick for the port's GFLOP/s)
    nd = 4096
    ga, gb = np.random.RandomState(0).rand(nd, nd), np.random.RandomState(1).rand(nd, nd)
    ga @ gb
    t_dgemm, _ = _median_time(lambda: ga @ gb, 3)
    host_dgemm_gflops = 2.0 * nd ** 3 / t_dgemm / 1e9
    del ga, gb

    # NumPy's element-wise passes over the N x M blocks are single-threaded; to give the CPU all of its cores the rows are
    # cut into shards evaluated by a thread pool (NumPy and BLAS release the GIL; the statistic bundle is additive over
    # row shards -- the same property the multi-GPU path uses), each worker with its share of the BLAS threads.
    from concurrent.futures import ThreadPoolExecutor
    from threadpoolctl import threadpool_limits
    ncpu = os.cpu_count() or 1
    workers = max(1, min(64, ncpu // 2))
    blas_per_worker = max(1, ncpu // workers)
    pool = ThreadPoolExecutor(max_workers=workers)

    def run_b(ns):
        u = so.u_algebra(prm, prob)                               # replicated M x M algebra: all BLAS threads
        nsh = max(1, min(workers, ns // 64))
        cuts = [(ns * i) // nsh for i in range(nsh + 1)]

        def shard(i):
            return so.local_stats(prm, prob, u, [x[cuts[i]:cuts[i + 1]] for x in X], [y[cuts[i]:cuts[i + 1]] for y in Y])[0]
        with threadpool_limits(limits=blas_per_worker, user_api="blas"):
            parts = list(pool.map(shard, range(nsh)))
        total = parts[0]
        for part in parts[1:]:
            total = total + part
        return so.finish(prm, prob, u, total)

    run_b(ns1)                                                    # warm-up (BLAS thread pool, page faults)
    t1, _ = _median_time(lambda: run_b(ns1), reps_b)
    t2, ts2 = _median_time(lambda: run_b(ns2), reps_b)
    per_row = max((t2 - t1) / (T * (ns2 - ns1)), 0.0) if ns2 > ns1 else t2 / (T * ns2)
    fixed = max(t1 - per_row * T * ns1, 0.0)
    full_extrapolated = fixed + per_row * T * N
    # [r4] ONE TRUE FULL-SIZE STEP of baseline B (all N rows of every task), timed; the two-sample split above stays as a
    # cross-check (`extrapolated_s`).  Skipped only when the prediction exceeds the budget (tiny hosts).
    full_measured = None
    if N > ns2 and full_extrapolated <= args.cpu_full_budget:
        t0f = time.perf_counter()
        run_b(N)
        full_measured = time.perf_counter() - t0f
    full = full_measured if full_measured is not None else full_extrapolated
    flops_sample = 3.0 * T * ns2 * Q * M * M + 20.0 * Q * M ** 3
    flops_full = 3.0 * T * N * Q * M * M + 20.0 * Q * M ** 3
    res = {"cpu_baseline": {
        "value": 1.0 / full, "unit": "steps/s", "full_step_measured": full_measured is not None,
        "full_step_s": full_measured, "extrapolated_s": full_extrapolated,
        "full_step_gflops": (flops_full / full_measured / 1e9) if full_measured else None, "cores": min(ncpu, workers * blas_per_worker), "host_cores": ncpu, "kind": "port",
        "blas": blas, "threads": threads, "row_shard_workers": workers, "blas_threads_per_worker": blas_per_worker,
        "gflops": flops_sample / t2 / 1e9, "host_dgemm_gflops": host_dgemm_gflops,
        "frac_of_host_dgemm": flops_sample / t2 / 1e9 / host_dgemm_gflops,
        "fixed_s": fixed, "per_row_s": per_row, "reps": reps_b,
        "sample": "baseline B: oracle.svmogp_oracle u_algebra + local_stats + finish (NumPy + %s, fp64; rows sharded over %d "
                  "worker threads x %d BLAS threads) on the first %d and %d of %d rows of each of the %d tasks, M=%d, Q=%d: "
                  "median of %d steps = %.2f s and %.2f s -> %.3f s independent of the rows + %.3e s per row (cross-check: "
                  "fixed + per_row * %d rows); `value` = 1 / (ONE measured full-size step of all rows, full_step_s) when "
                  "full_step_measured.  The same host runs a plain %d^3 dgemm at %.0f GFLOP/s: the port reaches %.1f %% "
                  "of that (NumPy element-wise passes and small per-shard GEMMs), so GPU/B overstates the hardware ratio"
                  % (blas, workers, blas_per_worker, ns1, ns2, N, T, M, Q, reps_b, t1, t2, fixed, per_row, T * N, nd,
                     host_dgemm_gflops, 100.0 * flops_sample / t2 / 1e9 / host_dgemm_gflops),
        "sample_seconds_per_step": t2}}
    # ---- parity at the headline M: the engine on exactly the sampled rows vs the oracle ------------------------------
    # (the engine of a 1-GPU run holds all rows; row_end restricts the evaluation to the sample)
    want = run_b(ns2)
    got = eng.elbo_grad(row_end=[ns2] * T, **prm)
    worst, worst_key = 0.0, None
    for k in ("elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"):
        a, b = np.asarray(got[k], float), np.asarray(want[k], float)
        err = float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
        if err > worst:
            worst, worst_key = err, k
    res["parity_at_headline_M"] = {"max_rel_err": worst, "worst": worst_key, "rows_per_task": ns2, "M": M, "Q": Q,
                                   "tolerance": 1e-5}
    if not (worst <= 1e-5):
        raise SystemExit("bench.py: engine and oracle disagree at the headline M: %s rel err %.3e" % (worst_key, worst))
    # ---- baseline A: the literal reference algorithm (N x N terms included) -------------------------------------------
    lit = {"kind": "port-literal", "cores": os.cpu_count(), "blas": blas, "threads": threads, "runs": []}

    def run_a(specs, prmA, Xa, Ya, Ma, Qa):
        pa = so.make_problem(specs, Qa, Ma, 1)
        r = so.inference_literal(prmA, pa, Xa, Ya, full_cov=True)
        so.assemble_literal(prmA, pa, Xa, r["grads"])

    from hetmogp_amd.synthetic import make_case
    c1 = [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})]
    p1, X1, Y1 = make_case(c1, [1000] * 3, M=50, Q=2, P=1, seed=20260930)
    run_a(c1, p1, X1, Y1, 50, 2)
    tA, _ = _median_time(lambda: run_a(c1, p1, X1, Y1, 50, 2), 3)
    lit["runs"].append({"config": "C1: T=3 [HetGaussian,Bernoulli,Categorical(3)], N_t=1000, M=50, Q=2", "seconds_per_step": tA,
                        "reps": 3})
    budget, spent, nt = float(args.cpu_literal_budget), 0.0, 2048
    while nt <= 8192 and N >= nt:
        Xa, Ya = [x[:nt] for x in X], [y[:nt] for y in Y]
        t0 = time.perf_counter()
        run_a(SPECS, prm, Xa, Ya, M, Q)
        dt = time.perf_counter() - t0
        spent += dt
        lit["runs"].append({"config": "headline mix, N_t=%d, M=%d, Q=%d" % (nt, M, Q), "seconds_per_step": dt, "reps": 1})
        if spent + 5.0 * dt > budget:                      # the next size (4x the rows) costs 4-5x (O(N^2) terms)
            break
        nt *= 4
    last = lit["runs"][-1]
    lit["value"] = 1.0 / last["seconds_per_step"]
    lit["unit"] = "steps/s at the size of the last run (O(N^2): not runnable at the full batch)"
    res["cpu_baseline_literal"] = lit
    pool.shutdown()
    return res


def _dominant(cat):
    k = max((k for k in cat if k not in ("total", "exchange")), key=lambda k: cat[k])
    return k, cat[k]


def _config_roofline(rows, Q, M, cat, tag):
    """FP64-MFMA roofline of a configuration's dominant contraction from the HIP-event spans of this run: the forward
    P~ = K^ C_q (2 rows Q M^2 algorithmic flops per launch, one launch per step) unless the weighted Gram takes longer
    (rows Q M^2, lower tiles).  `rocprof` names the committed rocprofv3 summaries of the same workload
    (tools/profile_configs.sh -> tools/summarize_profile.py), whose average kernel durations must agree."""
    fwd, gram = cat.get("forward_gemm", 0.0), cat.get("gram_gemm", 0.0)
    if fwd <= 0.0 and gram <= 0.0:
        return None
    if fwd >= gram:
        kern, fl, t = "rowpass_gemm_kernel<1> / gemm_f64_kernel (forward P~ = K^ C_q)", 2.0 * rows * Q * M * M, fwd
    else:
        kern, fl, t = "rowpass_gemm_kernel<2> (weighted Gram, lower tiles)", 1.0 * rows * Q * M * M, gram
    ach = fl / t / 1e9
    out = {"kernel": kern, "bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": ach / PEAK_FP64_MFMA_TFLOPS, "avg_launch_ms": t, "source": "HIP events on the engine's stream (this run)",
           "traffic": None}
    if tag:
        rnd = next((r for r in ("r05", "r04") if os.path.exists(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (r, tag)))), "r05")
        base = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, tag))
        if os.path.exists(base + "_kernel_stats.csv"):
            out["rocprof"] = "profiles/%s_%s_kernel_stats.csv" % (rnd, tag)
        hbm = base + "_pmc_hbm.csv"
        if os.path.exists(hbm):
            import csv
            want = "rowpass_gemm_kernel<1>" if fwd >= gram else "rowpass_gemm_kernel<2>"
            rows_ = [r for r in csv.DictReader(open(hbm)) if r["kernel"].startswith(want)]
            if rows_:
                out["traffic"] = int(max(rows_, key=lambda r: int(r["hbm_bytes_per_launch"]))["hbm_bytes_per_launch"])
                out["traffic_source"] = "profiles/%s_%s_pmc_hbm.csv" % (rnd, tag)
    return out


# Dependent-phase floors of the small-model evaluation (us), derived in DESIGN.md 11e from the phase stamps of the kernels and the
# micro-benchmarks under tools/probes/: what each graph node costs when nothing but its longest dependent chain is left -- the
# M = 50 factorisation's 50 dependent column steps (~200 cycles each: LDS broadcast + rsqrt chain) and four dependent 50^3
# products at one wave per SIMD (135 cycles per FP64 MFMA) for u_small; two such products + the PCIe gather for finish_small; one
# HBM round trip + one 64-row tile for the row kernels; one special-function chain for the quadrature -- plus 1.5 us per
# dependent kernel boundary (MI355X_MICROARCH.md).
C1_FLOOR_US = {"u_small_kernel": 18.0, "finish_small_kernel": 14.0, "small_fwd_kernel": 5.0, "quad_multi_kernel": 6.0,
               "small_bwd_kernel": 6.0, "small_red_kernel": 2.0}
C1_NODE_BOUNDARY_US, C1_UPLOAD_US = 1.5, 3.0


def _c1_latency_roofline(wall_ms):
    """C1 is latency-bound (4-250 CUs busy for 5-50 us per node): its `roofline` is a LATENCY model -- the sum of the dependent-phase
    floors of its graph nodes against the kernel time rocprofv3 measured for the same evaluation (committed CSV; a replayed hipGraph
    reports no per-node device time to the process itself)."""
    import csv
    src = next((f for f in ("r05_C1_kernel_stats.csv", "r04_C1_kernel_stats_small_path_v3.csv")
                if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    if src is None:
        return None
    kern = {}
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", src))):
        name = r["Name"]
        short = next((k for k in C1_FLOOR_US if k in name), None)
        if short:
            kern[short] = kern.get(short, 0.0) + float(r["AverageNs"]) / 1e3
    if not kern:
        return None
    floor = sum(C1_FLOOR_US[k] for k in kern) + C1_NODE_BOUNDARY_US * (len(kern) + 1) + C1_UPLOAD_US
    meas = sum(kern.values())
    dom = max(kern, key=kern.get)
    #  everything below except `wall_us_this_run` is read from a COMMITTED rocprofv3 summary and from hand-derived floors,
    # not measured by this process: a kernel regression would not move it.  The object says so in three places; the C1 number
    # this run does measure is its wall time (ms_per_step of the entry, `wall_us_this_run` here).
    return {"kernel": "hipGraph of the small-model evaluation (%d kernel nodes + upload)" % len(kern), "bound": "latency",
            "measured_in_this_run": False, "roofline_from_profile": "profiles/" + src,
            "note": "achieved / frac / kernels_us come from the committed rocprofv3 summary named in roofline_from_profile and the "
                    "hand-derived floors of DESIGN.md 11e; only wall_us_this_run is measured here",
            "achieved": meas, "peak": floor, "unit": "us of kernel time per evaluation (lower is better)", "frac": floor / meas,
            "kernels_us": {k: round(v, 2) for k, v in kern.items()}, "floors_us": C1_FLOOR_US, "dominant_kernel": dom,
            "dominant_kernel_us": round(kern[dom], 2), "wall_us_this_run": round(1e3 * wall_ms, 1),
            "source": "profiles/" + src, "traffic": None}


def _parity_c1_vs_reference_fixture():
    """C1 at its exact size against what the REFERENCE ITSELF returned (tests/golden/ref_c1_exact.npz: outputs of the reference's
    own SVMOGP.parameters_changed at N_t=1000, M=50, Q=2, captured in the authoring container): worst array-normalised error of
    ELBO + the 7 gradient arrays.  The fixture travels with the repository; the reference's Python does not."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "ref_c1_exact.npz")
    if not os.path.exists(path):
        return None
    from hetmogp_amd.engine import Engine
    g = np.load(path)
    specs = [(n, k) for n, k in json.loads(str(g["spec"]))]
    T, Q, M, P = int(g["T"]), int(g["Q"]), int(g["M"]), int(g["P"])
    e = Engine(specs, Q, M, P)
    e.set_data([g["Xbatch_%d" % t] for t in range(T)], [g["Ybatch_%d" % t] for t in range(T)])
    out = e.elbo_grad(Z=g["Z"], m_u=g["m_u"], L_flat=g["L_flat"], variance=g["variance"], lengthscale=g["lengthscale"],
                      W=g["W"], kappa=g["kappa"], W0=g["W0"], batch_scale=list(g["batch_scale"]))
    e.close()
    worst, wk = 0.0, None
    for k in ("elbo", "g_m_u", "g_L_u", "g_variance", "g_lengthscale", "g_W", "g_kappa", "g_Z"):
        a, b = np.asarray(out[k], float), np.asarray(g[k], float)
        err = float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
        if err > worst:
            worst, wk = err, k
    if not (worst <= 1e-5):
        raise SystemExit("bench.py: C1 disagrees with the reference-run fixture: %s rel err %.3e" % (wk, worst))
    return {"max_rel_err": worst, "worst": wk, "against": "tests/golden/ref_c1_exact.npz (the reference's own parameters_changed, "
            "N_t=1000, M=50, Q=2)", "tolerance": 1e-5, "reference_seconds_per_step_in_authoring_container": float(g["reference_seconds"])}


def _time_steps(eng, prm, steps, warmup=2, **kw):
    """Wall ms per `elbo_grad` (synchronous at return) and the per-family kernel ms of the last `steps` calls."""
    import gc
    gc.collect()              # (automatic collection is off in this process, see main(): collect between workloads instead)
    for _ in range(warmup):
        out = eng.elbo_grad(**dict(prm, **kw))
    t0, cat = time.perf_counter(), {}
    for _ in range(steps):
        out = eng.elbo_grad(**dict(prm, **kw))
        for k, v in eng.timings()[0].items():
            cat[k] = cat.get(k, 0.0) + v
    dt = 1e3 * (time.perf_counter() - t0) / steps
    return dt, {k: v / steps for k, v in cat.items()}, out


def other_configs(args):
    """The other BASELINE.json configurations on this GPU (a few steps each; synthetic inputs from the same generator,
    BASELINE.md 3).  `flops_executed` = 3 * rows * Q * M^2 (forward 2, lower-tile Gram 1) + 20 * Q * M^3 (replicated algebra,
    SURVEY 8d) per full-gradient step; `frac_of_peak` = that / ms_per_step / the FP64-MFMA peak."""
    import numpy as np
    from hetmogp_amd.engine import Engine
    from hetmogp_amd.synthetic import make_case
    K = max(1, args.other_steps)
    res = []

    def entry(name, rows, Q, M, ms, cat, out, tag=None, **extra):
        fl = 3.0 * rows * Q * M * M + 20.0 * Q * M ** 3
        dk, dms = _dominant(cat)
        e = {"workload": name, "ms_per_step": ms, "steps_per_s": 1e3 / ms, "flops_executed": fl,
             "tflops": fl / ms / 1e9, "frac_of_peak": fl / ms / 1e9 / PEAK_FP64_MFMA_TFLOPS,
             "dominant_kernel": dk, "dominant_kernel_ms": dms,
             "kernel_ms_per_step": {k: round(v, 4) for k, v in cat.items()}, "elbo": out["elbo"], "steps": K}
        e["roofline"] = _config_roofline(rows, Q, M, cat, tag)
        e.update(extra)
        res.append(e)

    def run(name, specs, N, M, Q, P, seed, steps=None, warmup=2, **extra):
        prm, X, Y = make_case(specs, [N] * len(specs), M=M, Q=Q, P=P, seed=seed)
        eng = Engine(specs, Q, M, P, reuse_outputs=True)
        eng.set_data(X, Y)
        ms, cat, out = _time_steps(eng, prm, steps or K, warmup=warmup)
        if not np.isfinite(out["elbo"]):
            raise SystemExit("bench.py: non-finite ELBO in " + name)
        return eng, prm, X, Y, ms, cat, out


    # C1 -- the reference's own CPU-runnable case (README usage snippet's likelihood list)
    c1 = [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})]
    # (a sub-millisecond step: 1000 timed evaluations behind 500 warm-ups -- the first ~0.5 s after idle run ~30 % slower)
    eng, prm, X, Y, ms, cat, out = run("C1", c1, 1000, 50, 2, 1, 20260930, steps=1000, warmup=500)
    cap, rep = eng.graph_stats()
    entry("C1: T=3 [HetGaussian,Bernoulli,Categorical(3)] Df=5, N_t=1000, M=50, Q=2, full-batch ELBO+gradients", 3000, 2, 50,
          ms, cat, out, steps_timed=1000, hipgraph_captures=cap, hipgraph_replays=rep,
          note="small-model path: fused LDS kernels (small_model.hip) replayed from a captured hipGraph, 17 kernels per evaluation; "
               "compare cpu_baseline_literal.runs[0]",
          parity_vs_reference_run=_parity_c1_vs_reference_fixture())
    lat = _c1_latency_roofline(ms)
    if lat is not None:          # (small-problem mode records no per-family spans: the table comes from the committed rocprof CSV)
        res[-1]["roofline"] = lat
        res[-1]["dominant_kernel"], res[-1]["dominant_kernel_ms"] = lat["dominant_kernel"], lat["dominant_kernel_us"] / 1e3
        res[-1]["kernel_ms_per_step"] = {k: round(v / 1e3, 5) for k, v in lat["kernels_us"].items()}
        res[-1]["kernel_ms_source"] = lat["roofline_from_profile"] + " (committed profile, NOT this run)"
    eng.close()
    # C2 -- the headline mix at M = 512
    eng, prm, X, Y, ms, cat, out = run("C2", SPECS, 200000, 512, 3, 1, 20260931)
    entry("C2: T=4 [Gaussian,Bernoulli,Poisson,Gamma] Df=5, N_t=200000, M=512, Q=3, full-batch ELBO+gradients", 800000, 3, 512,
          ms, cat, out, tag="C2", forward_tflops=2.0 * 800000 * 3 * 512 ** 2 / cat["forward_gemm"] / 1e9,
          gram_tflops=1.0 * 800000 * 3 * 512 ** 2 / cat["gram_gemm"] / 1e9)
    eng.close()
    # C3 -- SVI streaming: N_all = 1M rows per task, contiguous minibatches of 8192 rows per task and step
    res.append(svi_config(args, K))
    res.append(vem_c1_notebook(args))
    # C4 -- the share of ONE of 8 ranks: 125 000 of 1M rows of each of the 8 tasks, Q = 4, Df = 14
    c4 = [("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {}), ("Gaussian", {"sigma": 0.5}),
          ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
    eng, prm, X, Y, ms, cat, out = run("C4", c4, 125000, 1024, 4, 1, 20260933)
    entry("C4 (1/8 row share of one rank): T=8 [HetGaussian,Categorical(5),Beta,Exponential,Gaussian,Bernoulli,Poisson,Gamma] "
          "Df=14, 125000 of N_t=1M rows per task, M=1024, Q=4", 8 * 125000, 4, 1024, ms, cat, out,
          note="single-GPU measurement of a rank's share; the 8-GPU step adds one 16.9 MB all-reduce")
    eng.close()
    # C4 at its FULL size on ONE GPU: 8 tasks x 1M rows streamed in eight pools of 2^20 rows (the K^ / P~ workspaces of a pool are
    # 2 x 34 GB) -- the single-GPU denominator of the 8-GPU claim
    eng, prm, X, Y, ms, cat, out = run("C4F", c4, 1000000, 1024, 4, 1, 20260933, steps=min(K, 3), warmup=1)
    entry("C4 FULL SIZE on one GPU: T=8 [HetGaussian,Categorical(5),Beta,Exponential,Gaussian,Bernoulli,Poisson,Gamma] Df=14, "
          "N_t=1M rows per task (8M rows, 8 pools), M=1024, Q=4", 8 * 1000000, 4, 1024, ms, cat, out,
          note="denominator of the 8-GPU strong-scaling claim for config 4", steps_timed=min(K, 3))
    eng.close()
    del X, Y
    # C5 -- 2-D spatial, M = 2048, + predict_f on a 256 x 256 grid
    c5 = [("Categorical", {"K": 4}), ("Gaussian", {"sigma": 0.5})]
    eng, prm, X, Y, ms, cat, out = run("C5", c5, 50000, 2048, 2, 2, 20260934)
    g = np.stack(np.meshgrid(np.linspace(0, 1, 256), np.linspace(0, 1, 256), indexing="ij"), -1).reshape(-1, 2)
    eng.predict_f(g)                                   # sizes the row workspaces for the grid
    t0 = time.perf_counter()
    for _ in range(3):
        eng.predict_f(g)
    pms = 1e3 * (time.perf_counter() - t0) / 3
    entry("C5: T=2 [Categorical(4),Gaussian] Df=4, P=2, N_t=50000, M=2048, Q=2, full-batch ELBO+gradients", 100000, 2, 2048,
          ms, cat, out, tag="C5", predict_f_grid_ms=pms, predict_f_points=65536,
          predict_f_tflops=1.0 * 65536 * 2 * 2048 ** 2 / pms / 1e9)   # triangular fold: n Q M^2 executed
    eng.close()
    # HD -- the headline shape with a DENSE-VALUED operand: lengthscale = 40 inducing spacings (K^ has no exact zeros: the
    # headline's own K^ is > 90 % exact 0.0 because its lengthscales are about one spacing), GPy's jitter rung 4 forced so that
    # K_uu factorises.  Same kernels, same launch shapes, same flops: the FP64-MFMA rate on non-zero operands on record.
    prm, X, Y = make_case(SPECS, [200000] * 4, M=1024, Q=3, P=1, seed=20260929)
    prm["lengthscale"] = np.full(3, 40.0 / 1023.0)
    eng = Engine(SPECS, 3, 1024, 1, reuse_outputs=True)
    eng.set_data(X, Y)
    ms, cat, out = _time_steps(eng, prm, K, forced_rung=[4, 4, 4])
    entry("HD: headline shape (N_t=200000, M=1024, Q=3) with a dense-valued K^ (lengthscale = 40 spacings, jitter rung 4 forced)",
          800000, 3, 1024, ms, cat, out, tag="HD", finite=bool(np.isfinite(out["elbo"])),
          note="timing record only: K_uu at this lengthscale is numerically singular, the ELBO is not a parity quantity")
    eng.close()
    # HS / HSE -- the headline workload in the STRICT q(f) mode (HMOGP_CFG_STRICT_QF, DESIGN 6a / 13: the reference's solve-based forms;
    # what parity in the jitter-ladder regime costs).  Reported beside `value`, never mixed into it.  HS = a full-gradient evaluation
    # (two-solve form: A = dpotrs on the n x M side), HSE = an evaluation that asks for the q(u) gradients only (one-solve form).
    # [r6] per-kernel `roofline` entries: engine timing categories 9 / 10 separate the triangular solves and the
    # strict row statistics from the products (HIP events of this run).
    prm, X, Y = make_case(SPECS, [200000] * 4, M=1024, Q=3, P=1, seed=20260929)
    eng = Engine(SPECS, 3, 1024, 1, reuse_outputs=True, strict_qf=True)
    eng.set_data(X, Y)
    nqm2 = 800000.0 * 3 * 1024 ** 2
    from hetmogp_amd import _lib as _hl

    def strict_entry(tag, name, mask, solves, flops_rows, note, prm=prm, **kw):
        ms, cat, out = _time_steps(eng, prm, min(K, 3), warmup=1, group_mask=mask, **kw)
        fl = flops_rows * nqm2 + 20.0 * 3 * 1024 ** 3

        def rl(kernel, bound, work, unit_ms, peak, unit):
            ach = work / (unit_ms / 1e3) / (1e12 if unit == "TFLOP/s" else 1e9) if unit_ms > 0 else 0.0
            return {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                    "ms_per_step": unit_ms, "traffic": None}
        prod = (1.0 + (2.0 if mask != _hl.GROUP_QU else 0.0)) * nqm2          # T = A L_q (fold: n M^2) [+ P~ = A D: 2 n M^2]
        rls = [rl("trsm_panel_kernel<0|1> (%d blocked triangular solve(s): long-K update + in-tile substitution per 128-column block)"
                  % solves, "mfma", solves * nqm2, cat["trsm_solves"], PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"),
               rl("rowpass_fold_pair_kernel (T = A L_q, rowsum(T^2) in the epilogue)%s" %
                  (" + rowpass_gemm_kernel<1> (P~ = A (S Kuu^-1 - I))" if mask != _hl.GROUP_QU else ""), "mfma", prod,
                  cat["forward_gemm"], PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"),
               rl("rowpass_gemm_kernel<2> (Gram of A / of X, lower tiles)", "mfma", nqm2, cat["gram_gemm"], PEAK_FP64_MFMA_TFLOPS,
                  "TFLOP/s")]
        if cat.get("strict_rowstats", 0.0) > 0.5:     # (the E-step's only statistic kernel is a 0.04 ms combine: no roofline entry)
            stat_bytes = (8.0 * 800000 * 3 * 1024 * 2) if mask != _hl.GROUP_QU else 8.0 * 800000 * 3 * 10   # phase 1 streams K^ and P~
            rls.append(rl("strict_rowstats_kernel / trsm_stats_combine_kernel", "hbm", stat_bytes, cat["strict_rowstats"], PEAK_HBM_GBS,
                          "GB/s"))
        res.append({"workload": name, "ms_per_step": ms, "steps_per_s": 1e3 / ms, "flops_executed": fl, "tflops": fl / ms / 1e9,
                    "frac_of_peak": fl / ms / 1e9 / PEAK_FP64_MFMA_TFLOPS, "roofline": rls,
                    "kernel_ms_per_step": {k: round(v, 4) for k, v in cat.items()}, "elbo": out["elbo"], "steps": min(K, 3),
                    "cond_est": [float(c) for c in out["cond_est"]], "ill_conditioned": bool(out["ill_conditioned"]), "note": note})
    # HS: the headline parameters (lengthscale ~ inducing spacing, condition estimate 2-5).  HSL: the same shape where the strict mode is
    # actually NEEDED -- lengthscale = 4 x the inducing spacing, GPy's jitter rung 0 (forced, as the ladder would find it), estimate
    # 5.5e5.  Both take the one-solve form (estimate <= 1e6: DESIGN 13c).
    strict_entry("HS", "HS: headline workload in the strict q(f) mode (HMOGP_CFG_STRICT_QF), full gradients", _hl.GROUP_ALL, 1, 5.0,
                 "parity mode, not the headline.  One-solve form (round 6): 5 n Q M^2 contraction flops (ONE blocked triangular solve, "
                 "T = X W, P~ = X W2 with the phase-1 statistics in its epilogue, Gram of X); 298.7 ms in BENCH_r05 (two solves, "
                 "round-5 kernels)")
    prm_l = dict(prm, lengthscale=np.full(3, 4.0 / 1023.0))
    strict_entry("HSL", "HSL: headline shape at lengthscale = 4 x inducing spacing (jitter rung 0, cond(K_uu) ~ 1e7), strict q(f) mode, "
                 "full gradients", _hl.GROUP_ALL, 1, 5.0,
                 "the regime the strict mode exists for (the default path is 1e-4 ... 1e-3 off in g_W / g_Z / v_fd here): same kernels and "
                 "form as HS; against the reference-run fixture of this regime (lad_h_mix_M128_ladder) the evaluation is within 0.19 of the "
                 "element-wise 1e-5 criterion", prm=prm_l, forced_rung=[0, 0, 0])
    strict_entry("HSE", "HSE: headline workload in the strict q(f) mode, q(u) gradients only (an E-step: group_mask = QU)", _hl.GROUP_QU,
                 1, 3.0, "one-solve form (round 6): only the forward substitution X = K^ Luu^-T touches the n x M side; "
                 "3 n Q M^2 contraction flops (solve, T = X (Luu^-1 L_q), Gram of X)")
    eng.close()
    return res


def vem_c1_notebook(args):
    """[r6] VERDICT r5 item 2(i): the batch VEM driver a user of the reference runs (util.py:284-315: alternating L-BFGS-B over
    q(u) and over the hyper-parameters, <= 100 iterations each) on BASELINE config 1's shape with the NOTEBOOK's own
    hyper-parameters (demo.ipynb cell 7: lengthscale 0.05, variance 0.5, Z = linspace(0, 1, M)) -- l / h = 2.45, cond(K_uu) ~ 1e12:
    the regime in which GPy's jitchol decides the numbers.  Model built as the north-star spells it; strict_qf='auto' by default."""
    import warnings
    import numpy as np
    import hetmogp_amd as H
    from hetmogp_amd.kern import RBF
    from hetmogp_amd.synthetic import make_case
    c1 = [("HetGaussian", {}), ("Bernoulli", {}), ("Categorical", {"K": 3})]
    N, M, Q, P, iters = 1000, 50, 2, 1, 5
    prm, X, Y = make_case(c1, [N] * 3, M=M, Q=Q, P=P, seed=20260930)
    lik = H.HetLikelihood([H.HetGaussian(), H.Bernoulli(), H.Categorical(K=3)])
    np.random.seed(7)
    kern = [RBF(P, variance=0.5, lengthscale=0.05) for _ in range(Q)]
    Z = np.linspace(0.0, 1.0, M)[:, None]
    with warnings.catch_warnings(record=True) as wlog:
        warnings.simplefilter("always")
        guard = _StdoutToStderr()
        guard.__enter__()
        try:
            model = H.HetMOGP(X, [y.reshape(-1, 1) for y in Y], Z, kern, lik, lik.generate_metadata())
            e0 = float(model.log_likelihood()[0, 0])
            ev0, sev0 = model.evaluations, model.strict_evaluations
            t0 = time.perf_counter()
            H.vem_algorithm(model, stochastic=False, vem_iters=iters)
            wall = time.perf_counter() - t0
        finally:
            guard.__exit__()
    ev, sev = model.evaluations - ev0, model.strict_evaluations - sev0
    return {"workload": "T1: batch VEM trajectory, C1 shape [HetGaussian,Bernoulli,Categorical(3)], N_t=1000, M=50, Q=2 with the "
                        "notebook's hyper-parameters (lengthscale 0.05 on linspace(0,1,50), variance 0.5)",
            "driver": "hetmogp_amd.vem_algorithm(model, stochastic=False, vem_iters=5) on HetMOGP(X, Y, Z, kern_list, likelihood, "
                      "Y_metadata) -- util.py:284-315; constructor default strict_qf='auto'",
            "vem_iters": iters, "wall_s": wall, "evaluations": ev, "ms_per_evaluation": 1e3 * wall / max(ev, 1),
            "ms_per_step": 1e3 * wall / max(ev, 1), "steps_per_s": ev / wall,
            "strict_evaluations": sev, "strict_share": sev / float(max(ev, 1)), "strict_switches": int(model.strict_switches),
            "strict_now_at_end": bool(model._strict_now), "elbo_start": e0, "elbo_end": float(model.log_likelihood()[0, 0]),
            "cond_est_last": [float(c) for c in model.last["cond_est"]], "rungs_last": [int(r) for r in model.last["rungs"]],
            "ill_conditioned_warnings": len([w for w in wlog if "ill-conditioned" in str(w.message)]),
            "note": "evaluations = parameters_changed() calls of the L-BFGS-B line searches (a repeated evaluation at a default -> strict "
                    "switch counts once); strict evaluations run the regular kernels (solve-based forms), the others the fused "
                    "small-model path; lengthscale stays frozen in the first E-step only (util.py:285), so the optimiser may leave "
                    "the ill-conditioned regime by itself"}


def svi_config(args, K):
    """BASELINE config C3 through the facade (SVMOGP.stochastic_grad, svmogp.py:188-199: next contiguous minibatch, 4 E-steps
    then 1 M-step gating) with N_all = 1 000 000 rows per task resident in HBM, batch 8192 rows per task, q(u) and its
    Adadelta state device-resident (hmogp_qu_adadelta).  Reports the wall time of one training iteration (new batch +
    ELBO/gradients + optimiser update) and, separately, one full-gradient evaluation of a minibatch."""
    import numpy as np
    import hetmogp_amd as H
    from hetmogp_amd.engine import Engine
    from hetmogp_amd.kern import RBF
    from hetmogp_amd.synthetic import make_case
    import gc
    gc.collect()
    N_all, B, M, Q, P = 1000000, 8192, 1024, 3, 1
    prm, X, Y = make_case(SPECS, [N_all] * 4, M=M, Q=Q, P=P, seed=20260932)
    # (a) one full-gradient evaluation of a minibatch (all groups) straight through the C ABI
    eng = Engine(SPECS, Q, M, P, reuse_outputs=True)
    eng.set_data(X, Y)
    bs = [N_all / float(B)] * 4
    ms, cat, out = _time_steps(eng, prm, K, row_begin=[123456] * 4, row_end=[123456 + B] * 4, batch_scale=bs)
    eng.close()
    fl = 3.0 * 4 * B * Q * M * M + 20.0 * Q * M ** 3
    dk, dms = _dominant(cat)
    # (b) the training loop
    lik = H.HetLikelihood([H.Gaussian(sigma=0.5), H.Bernoulli(), H.Poisson(), H.Gamma()])
    np.random.seed(1)
    kern = [RBF(P, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
    model = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :P].copy(), kern_list=kern, likelihood=lik,
                     Y_metadata=lik.generate_metadata(), batch_size=B)
    model[".*.lengthscale"].fix()       # as util.vem_algorithm does (util.py:284-331)
    model[".*.kappa"].fix()
    model.Z.fix()
    model.stochastic = True
    opt = model.device_adadelta(step_rate=0.005, momentum=0.9)
    it = iter(opt)
    for _ in range(6):
        next(it)
    n_it = 5 * max(2, K)                # whole 4xE + 1xM cycles
    t0 = time.perf_counter()
    for _ in range(n_it):
        next(it)
    it_ms = 1e3 * (time.perf_counter() - t0) / n_it
    elbo = float(model._log_marginal_likelihood[0, 0])
    it.close()
    # (b2) the Adadelta loop once more with the opt-in exact-zero windows (exact_zero_windows="auto": sorted 1-D inputs): a
    # contiguous minibatch of sorted rows only touches the inducing points within ~38.6 lengthscales -- same results, fewer products
    np.random.seed(1)
    kernw = [RBF(P, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
    modelw = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :P].copy(), kern_list=kernw, likelihood=lik,
                      Y_metadata=lik.generate_metadata(), batch_size=B, exact_zero_windows="auto")
    modelw[".*.lengthscale"].fix()
    modelw[".*.kappa"].fix()
    modelw.Z.fix()
    modelw.stochastic = True
    itw = iter(modelw.device_adadelta(step_rate=0.005, momentum=0.9))
    for _ in range(6):
        next(itw)
    t0 = time.perf_counter()
    for _ in range(n_it):
        next(itw)
    w_ms = 1e3 * (time.perf_counter() - t0) / n_it
    w_elbo = float(modelw._log_marginal_likelihood[0, 0])
    w_on = bool(modelw.exact_zero_windows)
    itw.close()
    del modelw
    # (c) the same loop with NATURAL-gradient E-steps on the device-resident q(u) (hmogp_qu_natgrad; north-star)
    np.random.seed(1)
    kern2 = [RBF(P, variance=float(prm["variance"][q]), lengthscale=float(prm["lengthscale"][q])) for q in range(Q)]
    model2 = H.SVMOGP(X=X, Y=[y[:, None] for y in Y], Z=prm["Z"][:, :P].copy(), kern_list=kern2, likelihood=lik,
                      Y_metadata=lik.generate_metadata(), batch_size=B)
    model2[".*.lengthscale"].fix()
    model2[".*.kappa"].fix()
    model2.Z.fix()
    model2.stochastic = True
    ng = model2.device_natgrad(gamma=0.1, step_rate=0.005, momentum=0.9)
    it2 = iter(ng)
    for _ in range(35):                 # past the log-linear step-size warm-up (20 E-steps = 25 iterations)
        next(it2)
    t0 = time.perf_counter()
    for _ in range(n_it):
        next(it2)
    ng_ms = 1e3 * (time.perf_counter() - t0) / n_it
    ng_elbo = float(model2._log_marginal_likelihood[0, 0])
    it2.close()
    # (d) [r6] what a user of the reference's OWN driver sees: util.vem_algorithm(model, stochastic=True) -- the
    # façade's restatement of util.py:316-329: lengthscale and kappa frozen, variance / W free, Adadelta(step_rate 0.01, momentum
    # 0.9), 4 x E / 1 x M gating -- on a model built exactly as the north-star spells it (constructor default strict_qf="auto"),
    # for >= 200 iterations.  Z is frozen by hand (`model.Z.fix()`, as a user of the reference has to: its stochastic branch
    # never looks at optZ): with Z free the FIRST M-step of Adadelta moves every inducing point by 0.01 sqrt(1e-4 / 0.1) = 3.2e-4
    # = a third of the M = 1024 grid spacing in the direction of its gradient's sign, neighbouring inducing points collapse,
    # cond(K_uu) goes 5 -> 1e6 within five iterations and the ELBO to -inf (the reference's S_q = I start makes
    # K_uu^-1 S K_uu^-1 explode; same arithmetic in the reference): profiles/r06_svi_traj_probe.txt.
    #   traj       from lengthscale = 1.0 x inducing spacing (well conditioned: the default path throughout)
    #   traj_lad   from lengthscale = 4.0 x inducing spacing (cond(K_uu) ~ 1e7, GPy's jitter rung 0): what "auto" costs where it acts
    import warnings
    del model, model2
    gc.collect()
    h = 1.0 / (M - 1)

    def trajectory(ell_over_h, n_traj):
        np.random.seed(1)
        kern3 = [RBF(P, variance=float(prm["variance"][q]), lengthscale=ell_over_h * h) for q in range(Q)]
        model3 = H.HetMOGP(X, [y[:, None] for y in Y], prm["Z"][:, :P].copy(), kern3, lik, lik.generate_metadata(), batch_size=B)
        model3.Z.fix()
        ev0, sev0 = model3.evaluations, model3.strict_evaluations
        failed = None
        with warnings.catch_warnings(record=True) as wlog:
            warnings.simplefilter("always")
            guard = _StdoutToStderr()
            guard.__enter__()
            try:
                t0 = time.perf_counter()
                try:
                    H.vem_algorithm(model3, stochastic=True, vem_iters=n_traj - 1, verbose=False)   # (the callback stops at n_iter > vem_iters)
                except Exception as exc:                                                         # noqa: BLE001
                    failed = "%s: %s" % (type(exc).__name__, exc)
                traj_s = time.perf_counter() - t0
            finally:
                guard.__exit__()
        ev, sev = model3.evaluations - ev0, model3.strict_evaluations - sev0
        done = ev if failed else n_traj
        r = {"driver": "hetmogp_amd.vem_algorithm(model, stochastic=True, vem_iters=%d) on HetMOGP(X, Y, Z, kern_list, likelihood, "
                       "Y_metadata, batch_size=8192), model.Z.fix() -- util.py:316-329 / svmogp.py:168-217; constructor default "
                       "strict_qf='auto'" % (n_traj - 1),
             "start": "lengthscale = %.1f x inducing spacing (all latents), variance %s, S_q = I, m_q ~ 2.5 N(0,1) (svmogp.py:66-69)"
                      % (ell_over_h, [round(float(v), 3) for v in prm["variance"]]),
             "iterations": done, "failed": failed, "ms_per_iteration": 1e3 * traj_s / max(done, 1), "iterations_per_s": done / traj_s,
             "evaluations": ev, "strict_evaluations": sev, "strict_share": sev / float(max(ev, 1)),
             "strict_switches": int(model3.strict_switches), "strict_now_at_end": bool(model3._strict_now),
             "cond_est_last": [float(c) for c in model3.last["cond_est"]], "rungs_last": [int(r_) for r_ in model3.last["rungs"]],
             "elbo_first": float(model3.elbo[0, 0]), "elbo_last": float(model3.elbo[max(done - 1, 0), 0]),
             "ill_conditioned_warnings": len([w for w in wlog if "ill-conditioned" in str(w.message)]),
             "optimizer": "DeviceAdadelta (q(u) and its accumulators resident in HBM; iterates bit-identical to the host loop)"}
        del model3
        gc.collect()
        return r
    traj = trajectory(1.0, max(200, 5 * K))
    traj_lad = trajectory(4.0, 60)
    return {"svi_reference_driver_trajectory": traj, "svi_reference_driver_trajectory_ladder_regime": traj_lad,
            "workload": "C3: SVI streaming, T=4 [Gaussian,Bernoulli,Poisson,Gamma], N_all=1000000 rows/task resident, minibatch "
                        "8192 rows/task/step, M=1024, Q=3",
            "ms_per_step": ms, "steps_per_s": 1e3 / ms, "flops_executed": fl, "tflops": fl / ms / 1e9,
            "frac_of_peak": fl / ms / 1e9 / PEAK_FP64_MFMA_TFLOPS, "dominant_kernel": dk, "dominant_kernel_ms": dms,
            "kernel_ms_per_step": {k: round(v, 4) for k, v in cat.items()}, "elbo": out["elbo"], "steps": K,
            "roofline": _config_roofline(4 * B, Q, M, cat, "C3"),
            "note": "ms_per_step = one full-gradient evaluation of a minibatch (all parameter groups); svi_* = the facade's "
                    "training loop (4 E-steps with q(u) gradients only + 1 M-step, device-resident Adadelta)",
            "svi_ms_per_iteration": it_ms, "svi_iterations_per_s": 1e3 / it_ms, "svi_iterations_timed": n_it,
            "svi_elbo_last": elbo,
            "svi_exact_zero_windows_ms_per_iteration": w_ms, "svi_exact_zero_windows_on": w_on,
            "svi_exact_zero_windows_elbo_last": w_elbo,       # (same iterates as svi_elbo_last: only products with exact zeros are skipped)
            "svi_natgrad_ms_per_iteration": ng_ms, "svi_natgrad_elbo_last": ng_elbo, "svi_natgrad_gamma": ng.gamma_used,
            "svi_natgrad_rejected_steps": ng.rejected,
            "svi_natgrad_note": "same loop, E-steps = natural-gradient step of the device-resident q(u) (hmogp_qu_natgrad_async: "
                                "one M^3/3 factorisation of the reversed precision + one triangular inverse per step, committed on the device, "
                                "no host synchronisation and no host copy of q(u); E-step evaluations skip dL/dS L, HMOGP_EVAL_NO_G_L); rows "
                                "shuffled once, q(u) started at the prior, step size log-linear 1e-5 -> gamma over 20 E-steps "
                                "(DeviceNatGrad); elbo_last after 35 + timed iterations vs the Adadelta loop's after 6 + timed"}


if __name__ == "__main__":
    main()
