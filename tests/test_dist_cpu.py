"""CPU, world_size 2, gloo: the row-sharding + bundle all-reduce plumbing of hetmogp_amd/dist.py.  No GPU here, so
each rank's local row pass is played by the oracle's `local_stats` (the same additive bundle layout the HIP engine
produces) and `finish` by the oracle's replicated post-processing; what is under test is the host logic the product
uses: shard_ranges + all_reduce_host."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("hm_dist", os.path.join(ROOT, "hetmogp_amd", "dist.py"))
    hd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hd)                       # dist.py has no dependency on the HIP library
    from oracle import svmogp_oracle as so
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(path)
    prm, prob, X, Y, bs = so.load_case(g)
    u = so.u_algebra(prm, prob)
    rb, re = hd.shard_ranges([0] * prob["T"], [x.shape[0] for x in X], rank, world)
    Xs = [x[b:e] for x, b, e in zip(X, rb, re)]
    Ys = [y[b:e] for y, b, e in zip(Y, rb, re)]
    local, _ = so.local_stats(prm, prob, u, Xs, Ys, bs)
    total = hd.all_reduce_host(local)
    out = so.finish(prm, prob, u, total)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, float(out["elbo"]), out["g_Z"], out["g_L_u"], out["g_W"]))


def test_shard_rows_partition():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("hm_dist", os.path.join(ROOT, "hetmogp_amd", "dist.py"))
    hd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hd)
    for n in (0, 1, 7, 8, 200000, 1000003):
        for world in (1, 2, 3, 8):
            r = [hd.shard_rows(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_allreduce_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import svmogp_oracle as so
    path = os.path.join(ROOT, "tests", "golden", "inf_config2.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = np.load(path)
    prm, prob, X, Y, bs = so.load_case(g)
    want = so.elbo_grad_fused(prm, prob, X, Y, bs)
    for rank, elbo, gZ, gL, gW in res:
        assert abs(elbo - want["elbo"]) <= 1e-10 * abs(want["elbo"])
        for a, b in ((gZ, want["g_Z"]), (gL, want["g_L_u"]), (gW, want["g_W"])):
            assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b))
    assert abs(res[0][1] - res[1][1]) == 0.0 or abs(res[0][1] - res[1][1]) < 1e-9 * abs(res[0][1])


class _FakeEngine(object):
    """Host-only stand-in with the engine's exchange surface (wire buffer in NumPy) -- the negotiation logic of
    StatsReducer / attach_native_comm is what is under test, not the device."""

    def __init__(self, rank, fail_init_on=None):
        self.rank, self.fail_init_on = rank, fail_init_on
        self.wire = np.arange(10, dtype=np.float64) * (rank + 1)
        self.comm, self.destroyed, self.exchanged = (0, -1), 0, 0

    def wire_buffer(self):
        return 0, self.wire.size

    def wire_pack(self):
        pass

    def wire_unpack(self):
        pass

    def wire_read(self):
        return self.wire.copy()

    def wire_write(self, v):
        self.wire[...] = v

    def comm_info(self):
        return self.comm

    def comm_init(self, nranks, rank, uid):
        assert len(uid) == 128
        if self.fail_init_on == rank:
            raise RuntimeError("simulated ncclCommInitRank failure")
        self.comm = (nranks, rank)

    def comm_destroy(self):
        self.destroyed += 1
        self.comm = (0, -1)

    def step_exchange(self):
        self.exchanged += 1


def _negotiation_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("hm_dist", os.path.join(ROOT, "hetmogp_amd", "dist.py"))
    hd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hd)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    res["agree_all_true"] = hd.all_agree(True)
    res["agree_one_false"] = hd.all_agree(rank != 1)
    # default under gloo: host-staged all-reduce of the wire buffer
    e = _FakeEngine(rank)
    red = hd.StatsReducer(e)
    res["mode_default"] = red.mode
    red()
    res["wire_sum"] = e.wire.copy()
    api = (lambda: True, lambda: bytes(range(128)))
    # native, every rank succeeds: the engine keeps its communicator, the exchange is the engine's
    e2 = _FakeEngine(rank)
    red2 = hd.StatsReducer(e2, mode="native", native_api=api)
    red2()
    res["native_ok"] = (red2.mode, e2.comm, e2.exchanged, red2.owns_comm)
    red2.close()
    res["native_closed"] = (e2.comm, e2.destroyed)
    # native, rank 1 fails inside comm_init: EVERY rank must give up together, and the rank that succeeded drops its communicator
    e3 = _FakeEngine(rank, fail_init_on=1)
    try:
        hd.StatsReducer(e3, mode="native", native_api=api)
        res["native_partial"] = "no error"
    except RuntimeError as exc:
        res["native_partial"] = ("raised", e3.comm, e3.destroyed, "every rank" in str(exc))
    # library unavailable on one rank only
    api_half = (lambda: rank == 0, lambda: bytes(range(128)))
    e4 = _FakeEngine(rank)
    try:
        hd.StatsReducer(e4, mode="native", native_api=api_half)
        res["native_unavailable"] = "no error"
    except RuntimeError:
        res["native_unavailable"] = ("raised", e4.comm)
    # a communicator already attached on SOME ranks only -- every rank must issue the same collectives and reach
    # the same verdict (no hang, no mixed modes); the pre-attached communicator is left to its owner
    e5 = _FakeEngine(rank)
    if rank == 0:
        e5.comm = (world, 0)
    try:
        hd.StatsReducer(e5, mode="native", native_api=api)
        res["native_mixed"] = "no error"
    except RuntimeError as exc:
        res["native_mixed"] = ("raised", e5.comm, e5.destroyed, "some ranks only" in str(exc))
    # ... and attached on EVERY rank by the caller: taken as is, not owned
    e6 = _FakeEngine(rank)
    e6.comm = (world, rank)
    red6 = hd.StatsReducer(e6, mode="native", native_api=api)
    res["native_preattached"] = (red6.mode, red6.owns_comm)
    res["agree_min_max"] = hd.agree_min_max(rank + 3)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, res))


@pytest.mark.timeout(300)
def test_reducer_mode_negotiation_is_collective():
    """ADVICE r2 / VERDICT r2 item 4c: the exchange mode is agreed by a MIN all-reduce of a flag -- ranks either all take a mode
    or all give it up; a rank whose hmogp_comm_init succeeded while another's failed drops its communicator."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_negotiation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        r = res[rank]
        assert r["agree_all_true"] is True and r["agree_one_false"] is False
        assert r["mode_default"] == "host"
        assert np.array_equal(r["wire_sum"], np.arange(10) * 3.0)                 # (rank 0: x1) + (rank 1: x2)
        assert r["native_ok"] == ("native", (2, rank), 1, True)
        assert r["native_closed"] == ((0, -1), 1)
        assert r["native_partial"][0] == "raised" and r["native_partial"][1] == (0, -1) and r["native_partial"][3]
        assert r["native_unavailable"] == ("raised", (0, -1))
        assert r["native_mixed"] == ("raised", (2, 0) if rank == 0 else (0, -1), 0, True)
        assert r["native_preattached"] == ("native", False)
        assert r["agree_min_max"] == (3, 4)
    assert res[0]["native_partial"][2] == 1 and res[1]["native_partial"][2] == 0   # only the rank that had one destroyed it


def _dry_run_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("hm_dist", os.path.join(ROOT, "hetmogp_amd", "dist.py"))
    hd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hd)
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = {"n": 0}
    real = dist.all_reduce

    def counting(*a, **kw):
        calls["n"] += 1
        return real(*a, **kw)
    dist.all_reduce = counting
    res = {}
    res["plain_calls"] = (hd.all_agree(True), hd.agree_min_max(7), calls["n"])            # one rank: short-circuited, no collective
    hd.force_collectives(True)
    res["forced_calls"] = (hd.all_agree(True), hd.agree_min_max(7), calls["n"])           # forced: 1 + 2 flag reductions issued
    api = (lambda: True, lambda: bytes(range(128)))
    e = _FakeEngine(0)
    red = hd.StatsReducer(e, mode="native", native_api=api, single_rank_exchange=True)     # one-rank world negotiates like a larger one
    red()
    res["native"] = (red.mode, e.comm, e.exchanged)
    red.close()
    e2 = _FakeEngine(0)
    red2 = hd.StatsReducer(e2, single_rank_exchange=True)                                  # gloo: host mode, and it RUNS with one rank
    red2()
    res["host"] = (red2.mode, red2.n_calls, e2.wire.copy())
    e3 = _FakeEngine(0)
    red3 = hd.StatsReducer(e3)                                                             # default: a single rank skips the exchange
    red3()
    res["default_skips"] = (red3.mode, red3.n_calls)
    hd.force_collectives(False)
    dist.destroy_process_group()
    q.put(res)


@pytest.mark.timeout(300)
def test_one_rank_dry_run_switches():
    """bench.py --force-dist: `force_collectives` makes the negotiation's flag reductions real also in a world of one rank, and
    `StatsReducer(single_rank_exchange=True)` negotiates ("native" first) and exchanges like a larger world."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_dry_run_worker, args=(_free_port(), q))
    p.start()
    r = q.get(timeout=240)
    p.join(60)
    assert p.exitcode == 0
    assert r["plain_calls"] == (True, (7, 7), 0)
    assert r["forced_calls"] == (True, (7, 7), 3)
    assert r["native"] == ("native", (1, 0), 1)
    assert r["host"][0] == "host" and r["host"][1] == 1 and np.array_equal(r["host"][2], np.arange(10) * 1.0)
    assert r["default_skips"] == ("host", 0)


# ---- [r6] world size 8: the control flow of `bench.py --gpus 8` under gloo -------------------------------------
# bench.py's step closure, timed loop, max-over-ranks timing, per-rank gathers, exchange-mode sweep and teardown are functions of
# hetmogp_amd/dist.py (make_step / timed_steps / max_over_ranks / gather_floats / exchange_mode_sweep / teardown); bench.py calls them
# with the HIP engine and RCCL, this test with a stand-in engine whose row pass / finish are the oracle's and gloo.  What it pins:
# shard_ranges on C4's shape (8 tasks, 8 likelihood families, Q = 4; ragged tasks, one with FEWER rows than ranks), the
# collective-safe negotiation at 8 ranks, that every rank walks the same mode list, results == the single-process oracle on every
# rank, and the teardown order (library communicator -> barrier -> process group).  N > 1 RCCL itself has never executed anywhere.
C4_SPECS = [("HetGaussian", {}), ("Categorical", {"K": 5}), ("Beta", {}), ("Exponential", {}), ("Gaussian", {"sigma": 0.5}),
            ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
C4_ROWS = [13, 9, 8, 5, 16, 11, 21, 10]          # rows per task: none divisible by 8 but two, one shorter than the world


def _load_by_path(name, rel):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, *rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _OracleEngine(object):
    """The engine surface bench.py's distributed branch touches, played by the oracle (local_stats = the same additive bundle the HIP
    row pass produces, finish = the replicated post-processing).  The wire format here is the plain bundle."""

    def __init__(self, so, hd, prob, X, Y, rank, world, log):
        self.so, self.hd, self.prob, self.rank, self.world, self.log = so, hd, prob, rank, world, log
        self.X, self.Y = X, Y                      # THIS rank's rows only, as bench.py uploads them
        self.T, self.N = prob["T"], [x.shape[0] for x in X]
        self.comm, self.bundle, self.u, self.prm = (0, -1), None, None, None
        self._ms = {}

    # -- evaluation
    def step_begin(self, **prm):
        import time
        t0 = time.perf_counter()
        self.prm = prm
        self.u = self.so.u_algebra(prm, self.prob)
        t1 = time.perf_counter()
        self.bundle, _ = self.so.local_stats(prm, self.prob, self.u, self.X, self.Y, None)
        self._ms = {"mxm_algebra": 1e3 * (t1 - t0), "forward_gemm": 1e3 * (time.perf_counter() - t1), "exchange": 0.0}

    def step_finish(self, want_dL_dS=False):
        return self.so.finish(self.prm, self.prob, self.u, self.bundle)

    def step_exchange(self):                       # "native": stands in for pack -> ncclAllReduce -> unpack on the engine's stream
        import time
        assert self.comm == (self.world, self.rank), "exchange without a communicator spanning the group"
        t0 = time.perf_counter()
        self.bundle = self.hd.all_reduce_host(self.bundle)
        self._ms["exchange"] = 1e3 * (time.perf_counter() - t0)

    def elbo_grad(self, sharded=False, **prm):
        self.step_begin(**prm)
        if sharded:
            self.step_exchange()
        return self.step_finish()

    def timings(self):
        return dict(self._ms), {k: 1 for k in self._ms}

    # -- exchange surface
    def wire_buffer(self):
        return 0, self.so.stats_layout(self.prob)["size"]

    def wire_pack(self):
        pass

    def wire_unpack(self):
        pass

    def wire_read(self):
        return self.bundle.copy()

    def wire_write(self, v):
        self.bundle = np.array(v, dtype=float)

    def comm_info(self):
        return self.comm

    def comm_init(self, nranks, rank, uid):
        assert len(uid) == 128
        self.comm = (nranks, rank)
        self.log.append("comm_init")

    def comm_destroy(self):
        self.comm = (0, -1)
        self.log.append("comm_destroy")


def _world8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    hd = _load_by_path("hm_dist", ("hetmogp_amd", "dist.py"))
    syn = _load_by_path("hm_syn", ("hetmogp_amd", "synthetic.py"))
    from oracle import svmogp_oracle as so
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, Q, M, P = len(C4_SPECS), 4, 12, 1
    prm, X, Y = syn.make_case(C4_SPECS, C4_ROWS, M=M, Q=Q, P=P, seed=20260934)
    prob = so.make_problem(C4_SPECS, Q, M, P)
    rb, re = hd.shard_ranges([0] * T, C4_ROWS, rank, world)
    log = []
    eng = _OracleEngine(so, hd, prob, [x[b:e] for x, b, e in zip(X, rb, re)], [y[b:e] for y, b, e in zip(Y, rb, re)], rank, world, log)
    rows_rank = sum(e - b for b, e in zip(rb, re))
    api = (lambda: True, lambda: bytes(range(128)))
    reducer = hd.StatsReducer(eng)                                 # gloo: "host" is the only default candidate
    step = hd.make_step(eng, prm, reducer, True)

    def fence():
        dist.barrier()
    steps, warmup = 2, 1
    for _ in range(warmup):
        out = step()
    fence()
    elapsed, cat_ms, cat_n, walls, closing, out = hd.timed_steps(step, eng, steps, fence, reducer)
    mine = elapsed
    elapsed = hd.max_over_ranks(elapsed)
    fwd_rank = hd.gather_floats([cat_ms["forward_gemm"], float(cat_n["forward_gemm"]), float(rows_rank)], world)
    modes = hd.exchange_mode_sweep(eng, step, reducer, steps, warmup, fence, None, elapsed, cat_ms,
                                   make_reducer=lambda alt: hd.StatsReducer(eng, mode=alt, native_api=api))
    repl = [r[0] for r in hd.gather_floats([cat_ms["mxm_algebra"] / steps], world)]
    out_native = hd.make_step(eng, prm, None, False)()             # (plain single-rank call on this rank's rows: differs from the sum)
    log.append("before_teardown")
    hd.teardown(reducer)
    log.append("group_alive=%s" % dist.is_initialized())
    q.put((rank, dict(elbo=float(out["elbo"]), g_Z=np.asarray(out["g_Z"]), g_W=np.asarray(out["g_W"]), g_L_u=np.asarray(out["g_L_u"]),
                      rows=[int(r[2]) for r in fwd_rank], launches=[int(r[1]) for r in fwd_rank], modes=sorted(modes),
                      mode_ms=modes, repl_n=len(repl), elapsed_max=elapsed, elapsed_mine=mine, n_calls=reducer.n_calls,
                      local_only_elbo=float(out_native["elbo"]), log=log, rb=rb, re=re, n_walls=len(walls))))


@pytest.mark.timeout(600)
def test_world8_bench_control_flow():
    import torch.multiprocessing as mp
    syn = _load_by_path("hm_syn_parent", ("hetmogp_amd", "synthetic.py"))
    from oracle import svmogp_oracle as so
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    T, Q, M, P = len(C4_SPECS), 4, 12, 1
    prm, X, Y = syn.make_case(C4_SPECS, C4_ROWS, M=M, Q=Q, P=P, seed=20260934)
    want = so.elbo_grad_fused(prm, so.make_problem(C4_SPECS, Q, M, P), X, Y)
    # the shards tile every task exactly, also the 5-row task (three ranks hold none of its rows)
    for t in range(T):
        cuts = [(res[r]["rb"][t], res[r]["re"][t]) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == C4_ROWS[t] and all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
    assert sum(1 for r in range(world) if res[r]["re"][3] == res[r]["rb"][3]) == 3
    for r in range(world):
        o = res[r]
        assert abs(o["elbo"] - want["elbo"]) <= 1e-10 * abs(want["elbo"]), (r, o["elbo"], want["elbo"])
        for k in ("g_Z", "g_W", "g_L_u"):
            assert np.max(np.abs(o[k] - want[k])) <= 1e-9 * np.max(np.abs(want[k])), (r, k)
        assert o["rows"] == [sum(res[k]["re"][t] - res[k]["rb"][t] for t in range(T)) for k in range(world)] and sum(o["rows"]) == sum(C4_ROWS)
        assert o["launches"] == [2] * world and o["n_walls"] == 2 and o["repl_n"] == world
        assert o["modes"] == ["host", "native"]                       # same list on every rank: "device" is refused collectively under gloo
        assert o["elapsed_max"] >= o["elapsed_mine"] - 1e-12 and o["elapsed_max"] == res[0]["elapsed_max"]
        assert o["n_calls"] == 2                                      # the timed loop reset the counters after the warm-up
        assert abs(o["local_only_elbo"] - want["elbo"]) > 1e-6 * abs(want["elbo"])   # (a rank alone sees an eighth of the rows)
        # teardown order: the sweep's "native" communicator was created and destroyed by the sweep itself; nothing is destroyed after
        # the process group
        assert o["log"] == ["comm_init", "comm_destroy", "before_teardown", "group_alive=False"], o["log"]
        assert all(v["ms_per_step"] > 0 and v["exchange_ms_per_step"] >= 0 for v in o["mode_ms"].values())
    assert len({res[r]["elbo"] for r in range(world)}) == 1           # bit-identical replicas (same reduction order on every rank)
