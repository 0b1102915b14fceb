"""One-process-per-GPU plumbing for bench.py and the multi-rank tests.

The DP hot path shards by independent tasks: every rank aligns its own
slice of the batch and there is NO data-path collective.  torch.distributed (backend
"nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the barriers around the
timed region and for the max/sum reductions of the timing scalars."""
import os
import time


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous slice [first, first+count) of `total` units owned by `rank`."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class Group:
    """Thin wrapper: becomes a no-op when world == 1 (no process group is created)."""

    def __init__(self, backend="nccl", device=None):
        self.rank, self.local_rank, self.world = env_world()
        self.backend = backend
        self.device = device
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    import torch
                    kw["device_id"] = torch.device(device)
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self._dist = dist

    def _tensor(self, value, dtype):
        import torch
        dev = self.device if self.backend == "nccl" else "cpu"
        return torch.tensor([value], dtype=dtype, device=dev)

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = self._tensor(float(value), torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self._dist is None:
            return int(value)
        import torch
        t = self._tensor(int(value), torch.int64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return int(t.item())

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()


def timed_steps(group, step_fn, sync_fn, steps, warmup):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device
    sync on both sides; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    group.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    group.barrier()
    sync_fn()
    return group.max(time.perf_counter() - t0)
