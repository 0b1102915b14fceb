"""One-process-per-GPU plumbing for bench.py and the multi-rank tests.

The DP hot path shards by independent tasks: no collective inside a matrix.  The
work split is: rank 0 owns the task list -- and, for real inputs, the sequences: it BROADCASTS them
as one packed pool (broadcast_pool) --, partitions the list by longest-processing-time-first over the
task costs (csadp_partition_lpt) and BROADCASTS the assignment; every rank aligns its part; fixed
16-byte result records (task id, DP score, consensus, FNV-1a of the two rows) are ALL-GATHERED, so
every rank holds the whole batch's outcome; and the aligned ROWS -- what the reference's consumer
prints (alignment.c:134-156, segment->alignedstrings of dynamicprogramming.c:1160) -- are GATHERED
to rank 0 as one padded byte buffer per rank (gather_rows).
torch.distributed: backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests.  Also the
barriers around the timed region and the max/sum reductions of the timing scalars."""
import os
import subprocess
import sys
import time

RECORD_INTS = 4          # one result record = 4 x int32 = 16 bytes: task id, score, consensus, fnv1a


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total, rank, world):
    """Contiguous slice [first, first+count) of `total` units owned by `rank`."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class Group:
    """Thin wrapper: becomes a no-op when world == 1 (no process group is created), unless
    CSADP_DIST_FORCE_GROUP=1 asks for a real one-rank group -- that is how the RCCL calls below are
    exercised on a one-GPU box."""

    def __init__(self, backend="nccl", device=None):
        self.rank, self.local_rank, self.world = env_world()
        self.backend = backend
        self.device = device
        self._dist = None
        if self.world > 1 or os.environ.get("CSADP_DIST_FORCE_GROUP") == "1":
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:          # only the forced one-rank group gets here
                import socket
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
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

    def _dev(self):
        return self.device if self.backend == "nccl" else "cpu"

    def broadcast_ints(self, values, count):
        """Rank 0's list of `count` ints on every rank (one broadcast)."""
        if self._dist is None:
            return list(values)
        import torch
        t = torch.tensor(list(values) if self.rank == 0 else [0] * count, dtype=torch.int32, device=self._dev())
        self._dist.broadcast(t, src=0)
        return t.cpu().tolist()

    def all_gather_records(self, records):
        """records: this rank's list of RECORD_INTS-tuples (ints that fit 32 bits).  Returns the
        records of ALL ranks, rank order (one max-reduce for the padding, one all-gather)."""
        if self._dist is None:
            return [tuple(r) for r in records]
        import torch
        most = int(self.max(len(records)))
        pad = torch.full((most + 1, RECORD_INTS), -1, dtype=torch.int32)
        pad[0, 0] = len(records)
        if records:
            pad[1:len(records) + 1] = torch.tensor([[_i32(v) for v in r] for r in records], dtype=torch.int32)
        pad = pad.to(self._dev())
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self._dist.all_gather(out, pad)
        res = []
        for t in out:
            t = t.cpu()
            n = int(t[0, 0])
            res.extend(tuple(int(v) for v in row) for row in t[1:n + 1].tolist())
        return res


    def broadcast_bytes(self, data):
        """Rank 0's bytes object on every rank: its length first, then the payload (two broadcasts)."""
        if self._dist is None:
            return bytes(data)
        import torch
        n = self.broadcast_ints([len(data) if self.rank == 0 else 0], 1)[0]
        if self.rank == 0:
            t = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(self._dev())
        else:
            t = torch.empty(n, dtype=torch.uint8, device=self._dev())
        if n:
            self._dist.broadcast(t, src=0)
        return t.cpu().numpy().tobytes()

    def broadcast_pool(self, seqs, rotations=None):
        """The packed sequence pool of SURVEY 8(e): rank 0's sequences (bytes) and rotations on every rank.
        Layout: int32 count, count x int32 length, count x int32 rotation, then the letters back to back."""
        import struct
        blob = b""
        if self.rank == 0:
            rotations = list(rotations) if rotations is not None else [0] * len(seqs)
            blob = struct.pack("<i", len(seqs)) + struct.pack("<%di" % len(seqs), *[len(x) for x in seqs]) + \
                struct.pack("<%di" % len(seqs), *rotations) + b"".join(seqs)
        blob = self.broadcast_bytes(blob)
        (n,) = struct.unpack_from("<i", blob, 0)
        lens = struct.unpack_from("<%di" % n, blob, 4)
        rots = list(struct.unpack_from("<%di" % n, blob, 4 + 4 * n))
        out, off = [], 4 + 8 * n
        for ln in lens:
            out.append(blob[off:off + ln])
            off += ln
        return out, rots

    def gather_rows(self, ids, rows):
        """The aligned rows of this rank's tasks to rank 0: ids[i] is the global task id of rows[i] (a list of equal-length byte
        strings, one per sequence of the task).  Every rank packs [id, nseq, length, rows...] records into one byte buffer,
        padded to the longest buffer of any rank (one max-reduce), and ONE gather brings them to rank 0.  Returns
        {task id: [rows]} on rank 0 (complete), {} elsewhere."""
        import struct
        mine = bytearray()
        for t, rs in zip(ids, rows):
            mine += struct.pack("<iii", int(t), len(rs), len(rs[0]) if rs else 0)
            for r in rs:
                mine += r
        if self._dist is None:
            return _unpack_rows(bytes(mine))
        import torch
        most = int(self.max(len(mine)))
        buf = torch.zeros(most + 8, dtype=torch.uint8)
        buf[:8] = torch.frombuffer(bytearray(struct.pack("<q", len(mine))), dtype=torch.uint8)
        if mine:
            buf[8:8 + len(mine)] = torch.frombuffer(mine, dtype=torch.uint8)
        buf = buf.to(self._dev())
        dst = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
        self._dist.gather(buf, dst, dst=0)
        if self.rank != 0:
            return {}
        out = {}
        for t in dst:
            raw = t.cpu().numpy().tobytes()
            (n,) = struct.unpack_from("<q", raw, 0)
            out.update(_unpack_rows(raw[8:8 + n]))
        return out


def _unpack_rows(raw):
    import struct
    out, off = {}, 0
    while off < len(raw):
        t, nseq, ln = struct.unpack_from("<iii", raw, off)
        off += 12
        out[t] = [raw[off + i * ln:off + (i + 1) * ln] for i in range(nseq)]
        off += nseq * ln
    return out


def _i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def u32(v):
    return v & 0xFFFFFFFF


def lpt_assignment(group, costs):
    """Rank 0 partitions `costs` (known to rank 0 at least) over group.world parts with the library's
    LPT partitioner and broadcasts part-of-task; every rank returns the same list."""
    n = len(costs)
    part = None
    if group.rank == 0:
        import csa_amd
        part, _ = csa_amd.partition_lpt([int(c) for c in costs], group.world)
    return group.broadcast_ints(part, n)


def run_sharded(group, costs, align_mine):
    """The multi-GPU flow of one batch: LPT -> per-rank task lists -> align -> gather.
    align_mine(ids) aligns the global task ids this rank owns and returns one (score, consensus, fnv1a, rows) per id
    (rows = the aligned strings of the task).  Returns (records by task id -- on every rank --, this rank's ids, imbalance,
    rows by task id -- complete on rank 0, empty elsewhere)."""
    part = lpt_assignment(group, costs)
    mine = [t for t, p in enumerate(part) if p == group.rank]
    got = align_mine(mine)
    recs = group.all_gather_records([(t, g[0], g[1], g[2]) for t, g in zip(mine, got)])
    by_id = {r[0]: (r[1], r[2], u32(r[3])) for r in recs}
    rows = group.gather_rows(mine, [g[3] for g in got])
    load = [0] * group.world
    for t, p in enumerate(part):
        load[p] += int(costs[t])
    total = sum(load)
    imbalance = max(load) * group.world / total if total else 1.0
    return by_id, mine, imbalance, rows


def spawn_ranks(script, argv, world):
    """`python bench.py --gpus N` invoked plainly: start one fresh child per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in its environment) BEFORE this process has touched a GPU, pass rank 0's
    stdout through, return the worst exit code.  Never re-execs a process that holds a device."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll: when one rank dies (no device, an import error), the others would sit in init_process_group or a barrier until the
    # collective's time-out -- end them (they are this process' own fresh children) and report the first failure
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = 128 - code if code < 0 else code       # death by signal s -> 128 + s, like a shell
                for q in live:
                    q.terminate()
    return rc


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
