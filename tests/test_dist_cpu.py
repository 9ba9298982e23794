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
