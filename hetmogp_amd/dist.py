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


class StatsReducer(object):
    """All-reduce of one engine's statistic bundle in its wire format (lower triangles of the symmetric H_q only:
    12.7 MB instead of 25.2 MB at M=1024, Q=3).  mode "device": the engine's wire buffer is aliased as a torch CUDA
    tensor and reduced in place by RCCL (no host copy); mode "host": read -> CPU all-reduce -> write back (gloo).
    `last_ms` = wall milliseconds of the last exchange (pack + all-reduce + unpack), `n_calls` = exchanges so far."""

    def __init__(self, engine, device=0, mode=None, group=None):
        import torch
        import torch.distributed as dist
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if mode is None:
            mode = "device" if (dist.is_initialized() and dist.get_backend(group) == "nccl") else "host"
        self.mode = mode
        self.tensor = None
        self.last_ms, self.total_ms, self.n_calls = 0.0, 0.0, 0
        self.device = int(device)
        if mode == "device":
            # The engine allocates with the system HIP runtime, torch carries its own copy: aliasing the engine's buffer as
            # a torch tensor works on the configurations tested (see tests/test_dist_gpu.py), but it is not something either
            # library promises.  If it is refused, fall back to "staged": a torch-owned CUDA tensor filled through the host
            # (two 12.7 MB PCIe copies per step instead of none) -- slower, never wrong.
            try:
                ptr, n = engine.wire_buffer()
                self.tensor = torch.as_tensor(_DevicePtr(ptr, n), device="cuda:%d" % self.device)
                if self.tensor.device.index != self.device:
                    raise RuntimeError("wire buffer aliased on %s, engine on device %d" % (self.tensor.device, self.device))
            except Exception as exc:                                   # noqa: BLE001
                import warnings
                warnings.warn("StatsReducer: cannot alias the engine's wire buffer as a torch tensor (%s); using the "
                              "host-staged RCCL path" % (exc,))
                self.mode = "staged"
        if self.mode == "staged":
            _, n = engine.wire_buffer()
            self.tensor = torch.empty(n, dtype=torch.float64, device="cuda:%d" % self.device)

    def __call__(self):
        if self.world == 1:
            return
        import time
        import torch
        import torch.distributed as dist
        t0 = time.perf_counter()
        self.engine.wire_pack()                          # synchronous: the triangle is in the wire buffer at return
        if self.mode == "device":
            try:
                dist.all_reduce(self.tensor, op=dist.ReduceOp.SUM, group=self.group)
                torch.cuda.synchronize(self.tensor.device)   # the engine's own stream consumes it next
            except RuntimeError as exc:
                if self.n_calls > 0:
                    raise
                # refused on first use.  The decision must be the same on every rank: RCCL either accepts the aliased
                # buffer everywhere or nowhere (same binaries, same driver), so no negotiation is attempted here.
                import warnings
                warnings.warn("StatsReducer: RCCL refused the aliased wire buffer (%s); using the host-staged RCCL path" % (exc,))
                self.mode = "staged"
                self.tensor = torch.empty(self.tensor.numel(), dtype=torch.float64, device="cuda:%d" % self.device)
        if self.mode == "staged":
            self.tensor.copy_(torch.from_numpy(self.engine.wire_read()))
            dist.all_reduce(self.tensor, op=dist.ReduceOp.SUM, group=self.group)
            self.engine.wire_write(self.tensor.cpu().numpy())
        elif self.mode == "device":
            pass
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
    engine.step_begin(row_begin=rb, row_end=re, **params)
    reducer()
    return engine.step_finish(want_dL_dS=want_dL_dS)
