"""world_size-2 test of the one-process-per-GPU plumbing on CPU (gloo): sharding by
independent tasks, barrier + max-over-ranks timing, sum of per-rank work; no data-path
collective exists on this path."""
import json
import os
import socket
import subprocess
import sys

from csa_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %r)
from csa_amd import dist as cdist
from csa_amd.synth import config4_tasks
g = cdist.Group(backend="gloo")
first, count = cdist.shard_range(10, g.rank, g.world)
tasks = config4_tasks(first, count, 64)
work = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
calls = []
def step():
    calls.append(1)
    time.sleep(0.01 * (g.rank + 1))          # rank 1 is the slow one
elapsed = cdist.timed_steps(g, step, lambda: None, 5, 2)
total = g.sum(work)
print(json.dumps({"rank": g.rank, "first": first, "count": count, "elapsed": elapsed, "total": total,
                  "calls": len(calls), "work": work}), flush=True)
g.close()
''' % ROOT


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for first, count in spans:
                assert first == pos
                pos += count
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_two_rank_gloo_timing_and_sums(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err.decode()
        outs.append(json.loads(out.decode().strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert [o["first"] for o in outs] == [0, 5] and [o["count"] for o in outs] == [5, 5]
    assert outs[0]["total"] == outs[1]["total"] == outs[0]["work"] + outs[1]["work"]
    assert all(o["calls"] == 7 for o in outs)                       # 2 warm-up + 5 timed
    # the reported time is the MAX over ranks: both ranks see rank 1's >= 5 * 20 ms
    assert abs(outs[0]["elapsed"] - outs[1]["elapsed"]) < 1e-6
    assert outs[0]["elapsed"] >= 0.1
