"""Row sharding and the one exchange step of the path: every data-dependent quantity of the ELBO
and its gradients is a sum over rows, so each rank runs `step_begin` on its own contiguous row range of every
task, the additive statistic bundle is sum-all-reduced once, and `step_finish` (replicated M x M algebra) runs on
every rank.  torch.distributed is plumbing only: backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the
CPU tests of this module and as the host-staged fallback."""
import numpy as np


def shard_rows(n_rows, rank, world):
    """Contiguous, balanced [begin, end) of `n_rows` rows for `rank` of `world` (first `n_rows % world` ranks get one
    extra row).  Mirrors the reference's contiguous minibatch slices (util.py:52-72) at the rank level."""
    base, extra = divmod(int(n_rows), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_ranges(row_begin, row_end, rank, world):
    """Split every task's active range [row_begin[t], row_end[t]) across ranks."""
    rb, re = [], []
    for b, e in zip(row_begin, row_end):
        s0, s1 = shard_rows(e - b, rank, world)
        rb.append(b + s0)
        re.append(b + s1)
    return rb, re


def all_reduce_host(vec, group=None):
    """Sum-all-reduce a host float64 vector through torch.distributed (any backend that accepts CPU tensors)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.numpy()


class _DevicePtr(object):
    """Minimal __cuda_array_interface__ carrier so torch can alias engine-owned HBM without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _flag_device(group, device):
    """Where a negotiation flag has to live for the group's backend (RCCL only moves device memory)."""
    import torch
    import torch.distributed as dist
    return torch.device("cuda", int(device)) if dist.get_backend(group) == "nccl" else torch.device("cpu")


_FORCE_COLLECTIVES = False


def force_collectives(on=True):
    """Dry runs of the distributed path with ONE rank (bench.py --force-dist): issue the flag reductions of the negotiation
    also when the group has a single rank, so that the very calls a multi-GPU launch makes are executed."""
    global _FORCE_COLLECTIVES
    _FORCE_COLLECTIVES = bool(on)


def all_agree(ok, group=None, device=0):
    """True on every rank iff `ok` is true on every rank (a MIN all-reduce of one flag).  Every mode decision of the
    exchange step goes through this, so ranks can never end up in different modes."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE_COLLECTIVES):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_flag_device(group, device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()) == 1)


def agree_min_max(value, group=None, device=0):
    """(min, max) of an integer over the ranks of `group`: two all-reduces issued unconditionally by every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _FORCE_COLLECTIVES):
        return int(value), int(value)
    lo = torch.tensor([int(value)], dtype=torch.int32, device=_flag_device(group, device))
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return int(lo.item()), int(hi.item())


def attach_native_comm(engine, group=None, device=0, api=None):
    """Give `engine` its own RCCL communicator spanning the ranks of `group` (hmogp_comm_init): rank 0 draws the
    ncclUniqueId through the library, torch.distributed only carries those 128 bytes.  Collective; returns True on
    every rank or False on every rank (then no rank keeps a communicator).  `api` = (comm_available, comm_unique_id)
    replaces the library's entry points (tests of the negotiation without a GPU)."""
    import torch
    import torch.distributed as dist
    if api is None:
        from .engine import comm_available, comm_unique_id
    else:
        comm_available, comm_unique_id = api
    id_bytes = 128                                                     # HMOGP_COMM_ID_BYTES
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if not all_agree(comm_available(), group, device):
        return False
    dev = _flag_device(group, device)
    uid = torch.zeros(id_bytes, dtype=torch.uint8, device=dev)
    ok = True
    if rank == 0:
        try:
            uid = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).to(dev)
        except Exception:                                              # noqa: BLE001
            ok = False
    if not all_agree(ok, group, device):
        return False
    dist.broadcast(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    try:
        engine.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))
    except Exception as exc:                                           # noqa: BLE001
        import warnings
        warnings.warn("hetmogp_amd.dist: hmogp_comm_init failed on rank %d (%s)" % (rank, exc))
        ok = False
    if not all_agree(ok, group, device):
        if ok:
            engine.comm_destroy()
        return False
    return True


class StatsReducer(object):
    """The exchange step of one engine: sum-all-reduce of the statistic bundle in its wire format (lower triangles of the
    symmetric H_q only: 12.7 MB instead of 25.2 MB at M=1024, Q=3).  Modes, best first:

      "native"  the library's own RCCL communicator: pack -> ncclAllReduce -> unpack enqueued on the engine's stream, no
                host synchronisation, nothing of torch on the data path (hmogp_comm_init / hmogp_step_exchange);
      "device"  the engine's wire buffer aliased as a torch CUDA tensor, reduced in place by torch.distributed (RCCL);
      "staged"  a torch-owned CUDA tensor filled through the host (two PCIe copies per step) -- slower, never wrong;
      "host"    read -> CPU all-reduce -> write back (gloo; the CPU tests of this module).

    `mode=None` picks the best mode EVERY rank can do: each candidate is probed locally and the outcome is agreed on with
    a MIN all-reduce of a flag (`all_agree`), so ranks cannot diverge; nothing is decided by catching an exception inside
    the step.  `last_ms` = host wall milliseconds of the last exchange (for "native": the DEVICE time of pack + all-reduce + unpack,
    category "exchange" of Engine.timings(), also when the step went through `sharded_elbo_grad`'s fused call), `n_calls` = exchanges so far."""

    def __init__(self, engine, device=0, mode=None, group=None, native_api=None, single_rank_exchange=False):
        import torch
        import torch.distributed as dist
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = int(device)
        self.tensor = None
        self.last_ms, self.total_ms, self.n_calls = 0.0, 0.0, 0
        self.owns_comm = False
        # single_rank_exchange=True: a world of ONE rank still negotiates like a larger one ("native" first) and runs every
        # exchange for real (pack -> all-reduce over one rank -> unpack) -- the dry run of the distributed path on a 1-GPU box
        # (bench.py --force-dist); the default skips the exchange of a single rank, which has nothing to exchange
        self.single_rank_exchange = bool(single_rank_exchange)
        nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if mode is not None:
            candidates = [mode]
        elif not nccl:
            candidates = ["host"]
        else:       # (a single rank has nothing to exchange: no communicator is set up for it unless asked for)
            candidates = ["native", "device", "staged"] if (self.world > 1 or self.single_rank_exchange) else ["device", "staged"]
        self.mode = None
        for cand in candidates:
            if cand == "native":
                if not dist.is_initialized():
                    raise RuntimeError("StatsReducer(mode='native') needs an initialised torch.distributed group to carry "
                                       "the ncclUniqueId (or attach the communicator yourself: Engine.comm_init)")
                # Collective-safe: EVERY rank issues the same two flag reductions whatever its local state is,
                # and the branch is taken on the AGREED values only.  state 2 = a communicator spanning the group is already
                # attached, 1 = none attached (one can be created), 0 = something else (wrong size): not usable.
                n_attached = engine.comm_info()[0]
                state = 2 if n_attached == self.world else (1 if n_attached == 0 else 0)
                lo, hi = agree_min_max(state, group, self.device)
                if lo == 2:                                                  # the caller attached one on every rank
                    self.mode = "native"
                elif lo == 1 and hi == 1:                                    # nobody has one: create it (collective)
                    if attach_native_comm(engine, group, self.device, native_api):
                        self.owns_comm = True
                        self.mode = "native"
                elif mode == "native":                                       # mixed states: same verdict on every rank
                    raise RuntimeError("StatsReducer(mode='native'): communicators are attached on some ranks only "
                                       "(states %d..%d); attach on all ranks or on none" % (lo, hi))
            elif cand == "device":
                # The engine allocates with the HIP runtime of the process, torch wraps the pointer: aliasing works on the
                # configurations tested (tests/test_dist_gpu.py) but neither library promises it -- probe it with a real
                # all-reduce of the (zero-filled, not yet used) wire buffer and agree on the outcome.
                ok = True
                try:
                    ptr, n = engine.wire_buffer()
                    t = torch.as_tensor(_DevicePtr(ptr, n), device="cuda:%d" % self.device)
                    if t.device.index != self.device:
                        raise RuntimeError("wire buffer aliased on %s, engine on device %d" % (t.device, self.device))
                except Exception:                                      # noqa: BLE001
                    ok, t = False, None
                if all_agree(ok, group, self.device):
                    try:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                        torch.cuda.synchronize(t.device)
                    except Exception:                                  # noqa: BLE001
                        ok = False
                    if all_agree(ok, group, self.device):
                        self.tensor, self.mode = t, "device"
            elif cand == "staged":
                _, n = engine.wire_buffer()
                self.tensor = torch.empty(n, dtype=torch.float64, device="cuda:%d" % self.device)
                self.mode = "staged"
            elif cand == "host":
                self.mode = "host"
            else:
                raise ValueError("unknown StatsReducer mode %r" % (cand,))
            if self.mode is not None:
                break
        if self.mode is None:
            raise RuntimeError("StatsReducer: mode %r is not available on every rank" % (mode,))

    def close(self):
        if self.owns_comm:
            self.engine.comm_destroy()
            self.owns_comm = False

    def __call__(self):
        if self.world == 1 and self.mode != "native" and not self.single_rank_exchange:
            return                                        # (a one-rank communicator still runs its three launches: tests)
        import time
        import torch
        import torch.distributed as dist
        t0 = time.perf_counter()
        if self.mode == "native":
            self.engine.step_exchange()                  # enqueued on the engine's stream; step_finish follows on it
        else:
            self.engine.wire_pack()                      # synchronous: the triangle is in the wire buffer at return
            if self.mode == "device":
                dist.all_reduce(self.tensor, op=dist.ReduceOp.SUM, group=self.group)
                torch.cuda.synchronize(self.tensor.device)   # the engine's own stream consumes it next
            elif self.mode == "staged":
                self.tensor.copy_(torch.from_numpy(self.engine.wire_read()))
                dist.all_reduce(self.tensor, op=dist.ReduceOp.SUM, group=self.group)
                self.engine.wire_write(self.tensor.cpu().numpy())
            else:
                self.engine.wire_write(all_reduce_host(self.engine.wire_read(), self.group))
            self.engine.wire_unpack()
        self.last_ms = 1e3 * (time.perf_counter() - t0)
        self.total_ms += self.last_ms
        self.n_calls += 1


def sharded_elbo_grad(engine, reducer, rank, world, row_begin=None, row_end=None, want_dL_dS=False, **params):
    """One evaluation over `world` ranks: begin on this rank's rows -> all-reduce -> finish."""
    T = engine.T
    row_begin = [0] * T if row_begin is None else list(row_begin)
    row_end = list(engine.N) if row_end is None else list(row_end)
    rb, re = shard_ranges(row_begin, row_end, rank, world)
    if reducer is not None and reducer.mode == "native":
        # the library holds the communicator: one call = begin -> all-reduce on the engine's stream -> finish
        out = engine.elbo_grad(want_dL_dS=want_dL_dS, row_begin=rb, row_end=re, sharded=True, **params)
        reducer.last_ms = float(engine.timings()[0].get("exchange", 0.0))   # device time of pack + all-reduce + unpack
        reducer.total_ms += reducer.last_ms
        reducer.n_calls += 1
        return out
    engine.step_begin(row_begin=rb, row_end=re, **params)
    if reducer is not None:
        reducer()
    return engine.step_finish(want_dL_dS=want_dL_dS)


# ---- the control flow of `bench.py --gpus N` (kept here so that the CPU tests can run the SAME code under gloo with a stand-in engine:
# ---- tests/test_dist_cpu.py::test_world8_bench_control_flow -- N > 1 has never executed on hardware) -------------
def _tensor_device(device):
    import torch
    return torch.device("cpu") if device is None else torch.device("cuda", int(device))


def max_over_ranks(value, device=None):
    """MAX all-reduce of one float64 (the bench contract's max-over-ranks timing).  `device=None`: CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_tensor_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(values, world, device=None):
    """all_gather of a short float64 vector: returns a list (one entry per rank) of lists."""
    import torch
    import torch.distributed as dist
    dev = _tensor_device(device)
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return [[float(x) for x in g.cpu().tolist()] for g in got]


def make_step(eng, prm, reducer, distributed):
    """One bench step as a closure.  world > 1, default: the engine holds its own RCCL communicator ("native") and
    hmogp_elbo_grad_sharded IS the step (row pass -> pack / ncclAllReduce / unpack on the engine's stream -> replicated finish, one
    host sync at the end); the other modes go begin -> reducer -> finish."""
    def step(red=reducer):
        if not distributed:
            return eng.elbo_grad(**prm)
        if red.mode == "native":
            return eng.elbo_grad(sharded=True, **prm)
        eng.step_begin(**prm)
        red()
        return eng.step_finish()
    return step


def timed_steps(step, eng, steps, fence, reducer=None):
    """EXACTLY `steps` steps between two fences (barrier + device synchronisation, supplied by the caller); per-family kernel
    milliseconds (HIP-event spans of the engine) summed over them.  Returns (elapsed_s_this_rank, cat_ms, cat_n, step_walls,
    closing_fence_ms, last_out)."""
    import time
    if reducer is not None:
        reducer.total_ms, reducer.n_calls = 0.0, 0
    t0 = time.perf_counter()
    cat_ms, cat_n, step_walls, out = {}, {}, [], None
    for _ in range(steps):
        ts = time.perf_counter()
        out = step()
        step_walls.append(1e3 * (time.perf_counter() - ts))
        ms, nl = eng.timings()
        for k in ms:
            cat_ms[k] = cat_ms.get(k, 0.0) + ms[k]
            cat_n[k] = cat_n.get(k, 0) + nl[k]
    tf = time.perf_counter()
    fence()
    elapsed = time.perf_counter() - t0
    return elapsed, cat_ms, cat_n, step_walls, 1e3 * (time.perf_counter() - tf), out


def exchange_mode_sweep(eng, step, reducer, steps, warmup, fence, device, elapsed, cat_ms, alternatives=("native", "device"),
                        make_reducer=None):
    """N > 1: the same steps with the OTHER exchange modes every rank can do (reported, never `value`).  Every rank walks the same
    list in the same order; a mode that is not available on every rank is skipped on every rank (StatsReducer agrees collectively
    and raises RuntimeError everywhere).  Returns {mode: {"ms_per_step", "exchange_ms_per_step"}} including the default mode's entry
    computed from the main timed loop (`elapsed` = max over ranks, seconds; `cat_ms` = this rank's kernel-family sums)."""
    import time
    modes = {}
    native_ms = cat_ms.get("exchange", 0.0) / steps
    modes[reducer.mode] = {"ms_per_step": 1e3 * elapsed / steps,
                           "exchange_ms_per_step": native_ms if reducer.mode == "native" else reducer.total_ms / max(reducer.n_calls, 1)}
    for alt in alternatives:
        if alt in modes:
            continue
        try:
            red_alt = make_reducer(alt)
        except RuntimeError:            # not available on every rank (agreed collectively): nothing to time
            continue
        for _ in range(max(1, warmup)):
            step(red_alt)
        fence()
        t0, ex = time.perf_counter(), 0.0
        red_alt.total_ms, red_alt.n_calls = 0.0, 0
        for _ in range(steps):
            step(red_alt)
            ex += eng.timings()[0].get("exchange", 0.0)
        fence()
        el = max_over_ranks(time.perf_counter() - t0, device)
        exch = ex / steps if red_alt.mode == "native" else red_alt.total_ms / max(red_alt.n_calls, 1)
        modes[alt] = {"ms_per_step": 1e3 * el / steps, "exchange_ms_per_step": exch}
        if red_alt.owns_comm:           # (a communicator created for this pass only)
            red_alt.close()
    return modes


def teardown(reducer):
    """Teardown order of a distributed run: the library's own communicator first (ncclCommDestroy on every rank), then a barrier,
    then torch's process group."""
    import torch.distributed as dist
    if reducer is not None:
        reducer.close()
    dist.barrier()
    dist.destroy_process_group()
