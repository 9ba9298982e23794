"""GPU: the device-side plumbing of the one exchange step -- the engine's statistic bundle aliased as a torch tensor
(no copy) and reduced by RCCL -- on the single GPU that is available (world_size 1), plus a 2-process gloo run on the
same GPU that drives the real engine through the host-staged reducer."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from hetmogp_amd.synthetic import make_case
    specs = [("Gaussian", {"sigma": 0.5}), ("Bernoulli", {}), ("Poisson", {}), ("Gamma", {})]
    prm, X, Y = make_case(specs, [3000, 2000, 2500, 1500], M=64, Q=3, P=1, seed=3)
    return specs, prm, X, Y


def test_bundle_aliases_as_torch_tensor_and_rccl_allreduce():
    import torch
    import torch.distributed as dist
    from hetmogp_amd.engine import Engine
    from hetmogp_amd import dist as hd
    specs, prm, X, Y = _case()
    e = Engine(specs, 3, 64, 1, small_path=False)    # (a split / sharded step always takes the regular kernels: bit-identity with the
    e.set_data(X, Y)                                 #  plain call is asserted on those)
    full = e.elbo_grad(**prm)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        red = hd.StatsReducer(e, device=0)
        assert red.mode == "device" and red.tensor.is_cuda and red.tensor.dtype == torch.float64
        e.step_begin(**prm)
        e.wire_pack()                                                    # lower triangles of H_q -> wire buffer
        host = e.wire_read()
        assert red.tensor.numel() == host.size < e.stats_read().size
        assert np.array_equal(red.tensor.cpu().numpy(), host)            # same memory, no copy
        dist.all_reduce(red.tensor)                                      # RCCL on engine-owned HBM (sum over 1 rank)
        torch.cuda.synchronize()
        assert np.array_equal(e.wire_read(), host)
        out = hd.sharded_elbo_grad(e, red, 0, 1, **prm)
        for k in ("elbo", "g_m_u", "g_L_u", "g_Z", "g_W"):
            assert np.array_equal(np.asarray(out[k]), np.asarray(full[k])), k   # deterministic reductions: bit-identical
    finally:
        dist.destroy_process_group()


def test_native_exchange_one_rank_communicator():
    """The in-library exchange (hmogp_comm_init -> hmogp_elbo_grad_sharded = begin -> pack -> ncclAllReduce -> unpack -> finish, all on
    the engine's stream) with a communicator of ONE rank: the same three launches as on N ranks, results bit-identical to the
    plain call; the three-call form (step_begin / step_exchange / step_finish) too."""
    import torch
    import torch.distributed as dist
    from hetmogp_amd.engine import Engine, comm_available, comm_unique_id
    from hetmogp_amd import dist as hd
    from hetmogp_amd._lib import HetMOGPError
    assert comm_available()
    specs, prm, X, Y = _case()
    # (small_path=False: a SHARDED step never takes the fused small-model kernels -- its path must not depend on one rank's row
    #  count, ADVICE r4 -- so bit-identity with the plain call is asserted on the regular kernels; the small path's plain call
    #  agrees with them to rounding, checked at the end)
    e = Engine(specs, 3, 64, 1, small_path=False)
    e.set_data(X, Y)
    full = e.elbo_grad(**prm)
    assert e.timings()[0]["exchange"] == 0.0 and e.comm_info() == (0, -1)
    with pytest.raises(HetMOGPError):
        e.step_begin(**prm)
        e.step_exchange()                                                # no communicator yet
    e.comm_init(1, 0, comm_unique_id())                                  # no torch.distributed needed at all
    assert e.comm_info() == (1, 0)
    plain = e.elbo_grad(**prm)                                           # never a collective, communicator or not (ABI v5)
    assert e.timings()[0]["exchange"] == 0.0 and np.array_equal(plain["g_L_u"], full["g_L_u"])
    out = e.elbo_grad(sharded=True, **prm)
    ms, nl = e.timings()
    assert ms["exchange"] > 0.0 and nl["exchange"] == 3
    for k in ("elbo", "g_m_u", "g_L_u", "g_Z", "g_W", "g_variance", "g_lengthscale"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(full[k])), k
    e.step_begin(**prm)
    e.step_exchange()
    with pytest.raises(HetMOGPError):
        e.step_exchange()                                                # once per step
    out3 = e.step_finish()
    assert np.array_equal(out3["g_L_u"], full["g_L_u"]) and out3["elbo"] == full["elbo"]
    # a rank that cannot contribute aborts its communicator, so that its peers fail instead of blocking
    with pytest.raises(ValueError):
        e.elbo_grad(sharded=True, **dict(prm, row_begin=[0] * len(X), row_end=[x.shape[0] + 1 for x in X]))
    assert e.comm_info() == (0, -1)
    es = Engine(specs, 3, 64, 1)                                          # default: fused small-model kernels for the plain call ...
    es.set_data(X, Y)
    small = es.elbo_grad(**prm)
    es.comm_init(1, 0, comm_unique_id())
    shard = es.elbo_grad(sharded=True, **prm)                            # ... regular kernels for the sharded one
    for k in ("elbo", "g_m_u", "g_L_u", "g_Z", "g_W", "g_variance", "g_lengthscale"):
        a, b = np.asarray(shard[k], float), np.asarray(small[k], float)
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b)), k
        assert np.array_equal(np.asarray(shard[k]), np.asarray(full[k])), k
    es.close()
    with pytest.raises(HetMOGPError):
        e.elbo_grad(sharded=True, **prm)                                 # no communicator any more: E_STATE, not a hang
    e.comm_init(1, 0, comm_unique_id())
    e.comm_destroy()
    assert e.comm_info() == (0, -1)
    # the same through the reducer inside a (one-rank) torch.distributed group: the id travels by dist.broadcast
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        red = hd.StatsReducer(e, device=0, mode="native")
        assert red.mode == "native" and red.owns_comm and e.comm_info() == (1, 0)
        out = hd.sharded_elbo_grad(e, red, 0, 1, **prm)
        assert np.array_equal(out["g_L_u"], full["g_L_u"]) and e.timings()[0]["exchange"] > 0.0
        red.close()
        assert e.comm_info() == (0, -1)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from hetmogp_amd.engine import Engine
    from hetmogp_amd import dist as hd
    from test_dist_gpu import _case
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs, prm, X, Y = _case()
    e = Engine(specs, 3, 64, 1, device=0)                 # both ranks share the one GPU of the box
    e.set_data(X, Y)
    red = hd.StatsReducer(e, device=0)
    assert red.mode == "host"
    out = hd.sharded_elbo_grad(e, red, rank, world, **prm)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out["elbo"], out["g_Z"], out["g_L_u"]))


@pytest.mark.timeout(600)
def test_two_ranks_row_sharded_match_single_rank():
    import torch.multiprocessing as mp
    from hetmogp_amd.engine import Engine
    specs, prm, X, Y = _case()
    e = Engine(specs, 3, 64, 1)
    e.set_data(X, Y)
    full = e.elbo_grad(**prm)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, elbo, gZ, gL in res:
        assert abs(elbo - full["elbo"]) < 1e-10 * abs(full["elbo"])
        assert np.max(np.abs(gZ - full["g_Z"])) < 1e-9 * np.max(np.abs(full["g_Z"]))
        assert np.max(np.abs(gL - full["g_L_u"])) < 1e-9 * np.max(np.abs(full["g_L_u"]))
    assert res[0][1] == res[1][1]                          # replicated finish: identical on every rank


def _facade_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch.distributed as dist
    from test_facade_gpu import build_model
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_config2_full.npz"))
    import hetmogp_amd.svmogp as sv
    orig = sv.SVMOGP.__init__

    def patched(self, *a, **kw):                      # same fixture builder, sharded evaluation
        kw["distributed"] = True
        return orig(self, *a, **kw)
    sv.SVMOGP.__init__ = patched
    model = build_model(g)
    model.parameters_changed()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, float(model.log_likelihood()[0, 0]), np.asarray(model.Z.gradient), np.asarray(model.q_u_chols.gradient)))


@pytest.mark.timeout(600)
def test_facade_distributed_rows_match_reference_fixture():
    """SVMOGP(distributed=True) on two ranks (gloo, both on the one GPU): every rank reproduces the reference's
    parameters_changed() outputs of the fixture."""
    import torch.multiprocessing as mp
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_config2_full.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_facade_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, elbo, gZ, gL in res:
        ref = float(np.asarray(g["elbo"]).reshape(-1)[0])
        assert abs(elbo - ref) < 1e-8 * abs(ref)
        assert np.max(np.abs(gZ - g["g_Z"])) < 1e-8 * np.max(np.abs(g["g_Z"]))
        assert np.max(np.abs(gL - g["g_L_u"])) < 1e-8 * np.max(np.abs(g["g_L_u"]))


# ------------------------------------------------------------------------------------------------ >= 2 GPUs: real RCCL
def _nccl_worker(rank, world, port, q, mode=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from hetmogp_amd.engine import Engine
    from hetmogp_amd import dist as hd
    from test_dist_gpu import _case
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    specs, prm, X, Y = _case()
    rb, re = hd.shard_ranges([0] * len(X), [x.shape[0] for x in X], rank, world)
    e = Engine(specs, 3, 64, 1, device=rank)                                  # one GPU per rank
    e.set_data([x[b:e_] for x, b, e_ in zip(X, rb, re)], [y[b:e_] for y, b, e_ in zip(Y, rb, re)])   # only its rows
    red = hd.StatsReducer(e, device=rank, mode=mode)
    if mode in (None, "native"):                      # the default on >= 2 ranks: the library's own communicator
        assert red.mode == "native" and e.comm_info() == (world, rank)
    else:
        assert red.mode == mode and red.tensor.device.index == rank
    e.step_begin(**prm)
    red()
    out = e.step_finish()
    if red.mode == "native":                          # and the one-call form (hmogp_elbo_grad_sharded: collective)
        out1 = e.elbo_grad(sharded=True, **prm)
        assert out1["elbo"] == out["elbo"] and np.array_equal(out1["g_L_u"], out["g_L_u"])
        assert e.timings()[0]["exchange"] > 0.0
    ones = torch.ones(1, dtype=torch.float64, device="cuda")
    dist.all_reduce(ones)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out["elbo"], out["g_Z"], out["g_L_u"], float(ones.item()), red.last_ms))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", [None, "device", "staged"])
def test_two_ranks_two_gpus_rccl_match_single_rank(mode):
    """BASELINE config C4's mechanism on real hardware: one process per GPU, rows sharded, the wire bundle all-reduced by
    RCCL -- by the library's own communicator on the engine's stream (default, "native"), in place on engine-owned HBM
    through torch ("device"), or host-staged.  Skipped on boxes with a single GPU."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the nccl backend refuses two ranks on one device)")
    import torch.multiprocessing as mp
    from hetmogp_amd.engine import Engine
    specs, prm, X, Y = _case()
    e = Engine(specs, 3, 64, 1)
    e.set_data(X, Y)
    full = e.elbo_grad(**prm)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, elbo, gZ, gL, nranks, ms in res:
        assert nranks == 2.0
        assert abs(elbo - full["elbo"]) < 1e-10 * abs(full["elbo"])
        assert np.max(np.abs(gZ - full["g_Z"])) < 1e-9 * np.max(np.abs(full["g_Z"]))
        assert np.max(np.abs(gL - full["g_L_u"])) < 1e-9 * np.max(np.abs(full["g_L_u"]))
    assert res[0][1] == res[1][1]


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without a rendezvous in the environment re-executes itself under torch.distributed.run
    and prints one JSON line whose rccl_ranks == 2.  Skipped on boxes with a single GPU."""
    import json
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows",
                        "20000", "--inducing", "256"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and sum(line["rows_per_rank"]) == 4 * 20000
    assert line["allreduce_ms_per_step"] > 0.0 and line["reducer_mode"] == "native"
    assert "native" in line["exchange_modes_ms_per_step"]          # ("device" is timed too wherever torch can alias the buffer)
    assert len(line["replicated_ms_per_rank"]) == 2


@pytest.mark.timeout(900)
def test_bench_force_dist_one_rank_dry_run():
    """`python bench.py --gpus 1 --force-dist`: the whole distributed branch of bench.py (nccl process group, proof all-reduce,
    StatsReducer negotiation with its flag reductions, the library's own RCCL communicator, hmogp_elbo_grad_sharded as the step,
    the alternative-exchange-mode loop, gathers, teardown order) with a world of ONE rank -- so that the first real multi-GPU
    launch cannot die on Python.  Its line must agree with the plain single-GPU line of the same workload."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--gpus", "1", "--steps", "4", "--warmup", "2", "--rows", "50000", "--inducing", "512"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, env=env, capture_output=True,
                           text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]                      # the contract: ONE JSON line on stdout
        return json.loads(lines[0])

    forced = run(["--force-dist"])
    plain = run(["--no-other-configs", "--no-cpu-baseline", "--no-exact-zero-pass"])
    assert forced["force_dist"] is True and forced["n_gpus"] == 1 and forced["rccl_ranks"] == 1
    assert forced["reducer_mode"] == "native" and forced["allreduce_ms_per_step"] > 0.0
    assert set(forced["exchange_modes_ms_per_step"]) >= {"native"}
    assert forced["rows_per_rank"] == [4 * 50000] and len(forced["replicated_ms_per_rank"]) == 1
    assert forced["allreduce_bytes"] > 0
    # [r6] what the first run with N > 1 needs in its line without an edit: the dominant kernel's roofline on
    # every rank and the exchange / the replicated chain as shares of the step
    rp = forced["roofline_per_rank"]
    assert len(rp) == 1 and rp[0]["rank"] == 0 and rp[0]["rows"] == 4 * 50000 and rp[0]["launches"] == 4
    assert 0.3 < rp[0]["frac"] < 1.0 and rp[0]["bound"] == "mfma" and abs(rp[0]["frac"] - forced["roofline"]["frac"]) < 1e-9
    assert 0.0 < forced["allreduce_frac_of_step"] < 0.2 and 0.0 < forced["replicated_frac_of_step"] < 0.5
    assert forced["n_gt_1_rccl_executed_before_this_run"] is False
    assert abs(forced["elbo"] - plain["elbo"]) <= 1e-12 * abs(plain["elbo"])      # same numbers ...
    assert forced["ms_per_step"] < 1.25 * plain["ms_per_step"] + 0.5               # ... in the same time (+ one-rank exchange)
    print("force-dist %.3f ms/step (exchange %.3f) vs plain %.3f ms/step; modes %s" % (
        forced["ms_per_step"], forced["allreduce_ms_per_step"], plain["ms_per_step"], forced["exchange_modes_ms_per_step"]))


def _facade_nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from test_facade_gpu import build_model
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_config2_full.npz"))
    import hetmogp_amd.svmogp as sv
    orig = sv.SVMOGP.__init__

    def patched(self, *a, **kw):                      # same fixture builder; device defaults to LOCAL_RANK
        kw["distributed"] = True
        return orig(self, *a, **kw)
    sv.SVMOGP.__init__ = patched
    model = build_model(g)
    assert model._dist[1].mode == "native" and model._engine is not None
    model.parameters_changed()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, float(model.log_likelihood()[0, 0]), np.asarray(model.Z.gradient), np.asarray(model.q_u_chols.gradient)))


@pytest.mark.timeout(600)
def test_facade_distributed_two_gpus_rccl_matches_reference_fixture():
    """SVMOGP(distributed=True) on two ranks, one GPU each, backend nccl (device defaults to LOCAL_RANK, the library's
    own RCCL communicator): every rank reproduces the reference's parameters_changed() fixture.  Skipped on 1-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    import torch.multiprocessing as mp
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_config2_full.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_facade_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, elbo, gZ, gL in res:
        assert abs(elbo - float(np.ravel(g["elbo"])[0])) < 1e-8 * abs(float(np.ravel(g["elbo"])[0]))
        assert np.max(np.abs(gZ - g["g_Z"])) < 1e-8 * np.max(np.abs(g["g_Z"]))
        assert np.max(np.abs(gL - g["g_L_u"])) < 1e-8 * np.max(np.abs(g["g_L_u"]))


@pytest.mark.timeout(600)
def test_plain_c_two_ranks_file_rendezvous_no_torch(tmp_path):
    """The row-sharded step from PLAIN C, one process per GPU, no Python / torch / MPI in the ranks: examples/c_abi_two_rank.c
    (ncclUniqueId handed from rank 0 to rank 1 through a file; hmogp_comm_init; hmogp_elbo_grad_sharded).  Every rank must
    print the same ELBO / gradients as the unsharded single-GPU evaluation rank 0 runs afterwards.  Skipped on 1-GPU boxes."""
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    exe = str(tmp_path / "two_rank")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_two_rank.c"), "-o", exe,
                    "-L" + os.path.join(ROOT, "hetmogp_amd"), "-lhetmogp_hip", "-Wl,-rpath," + os.path.join(ROOT, "hetmogp_amd"),
                    "-lm"], check=True)
    idfile = str(tmp_path / "hm_id.bin")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(r), "2", idfile], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (so_, se_) in zip(procs, outs):
        assert p.returncode == 0, se_[-2000:]
    lines = [l for so_, _ in outs for l in so_.splitlines() if " elbo " in l]
    vals = {}
    for l in lines:
        tok = l.split()
        key = (tok[1], tok[2])                                  # ("0" | "1", "sharded" | "single")
        nums = []
        for x in tok:
            try:
                nums.append(float(x))                          # [rank, elbo, g_var x2, g_ell x2, three sums, exchange_ms]
            except ValueError:
                pass
        vals[key] = nums
    single = vals[("0", "single")]
    for r in ("0", "1"):
        got = vals[(r, "sharded")]
        for a, b in zip(got[1:9], single[1:9]):                 # elbo, g_var x2, g_ell x2, three weighted gradient sums
            assert abs(a - b) <= 1e-9 * max(1.0, abs(b)), (r, a, b)
    assert vals[("0", "sharded")][1:9] == vals[("1", "sharded")][1:9]          # replicated finish: identical bits on every rank


def _svi_worker(rank, world, port, q, optimizer):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch.distributed as dist
    import hetmogp_amd as H
    from test_facade_gpu import build_model
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hetmogp_amd.svmogp as sv
        orig = sv.SVMOGP.__init__

        def patched(self, *a, **kw):
            kw["distributed"] = True
            return orig(self, *a, **kw)
        sv.SVMOGP.__init__ = patched
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_h_mix_M128.npz"))
    model = build_model(g, batch_size=96)
    np.random.seed(0)
    H.vem_algorithm(model, stochastic=True, vem_iters=14, step_rate=0.01, qu_optimizer=optimizer, natgrad_gamma=0.05)
    res = (rank, model.elbo[:14, 0].copy(), np.asarray(model.q_u_means.values).copy(), np.asarray(model.q_u_chols.values).copy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    q.put(res)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("optimizer", ["adadelta", "natgrad"])
def test_device_resident_svi_loop_in_a_row_sharded_model(optimizer):
    """[r4] The SVI loops with q(u) resident in HBM (DeviceAdadelta / DeviceNatGrad) inside SVMOGP(distributed=True): two ranks
    (gloo, both on the one GPU) shard every minibatch's rows, all-reduce the bundle and apply the SAME device-side update to
    their replicas of q(u).  The ELBO trace and the final q(u) equal the single-process loop's (to the summation order of the
    sharded statistics) and are bit-identical on the two ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_svi_worker, args=(0, 1, _free_port(), q1, optimizer))
    p1.start()
    _, e1, m1, L1 = q1.get(timeout=400)
    p1.join(60)
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_svi_worker, args=(r, 2, port, q, optimizer)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, ea, ma, La), (_, eb, mb, Lb) = res
    assert np.array_equal(ea, eb) and np.array_equal(ma, mb) and np.array_equal(La, Lb)       # replicas never diverge
    assert np.all(np.isfinite(ea))
    assert np.max(np.abs(ea - e1)) < 1e-7 * np.max(np.abs(e1))
    assert np.max(np.abs(ma - m1)) < 1e-6 * np.max(np.abs(m1)) and np.max(np.abs(La - L1)) < 1e-6 * np.max(np.abs(L1))
