#!/usr/bin/env python3
"""Chunked launches (batches with jobs wider than four strips) with the partial last chunks and the narrow jobs sharing workgroups (CSADP_BITS_PACK=1) against one workgroup per chunk (=0)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
r = random.Random(7)
shapes = [("256 pairs of 500-40000", [r.randrange(500, 40000) for _ in range(256)]),
          ("128 pairs of 2000-60000", [r.randrange(2000, 60000) for _ in range(128)]),
          ("200 pairs: 100 of 30000, 100 of 5000", [30000] * 100 + [5000] * 100),
          ("64 pairs of 33000", [33000] * 64), ("40 pairs of 50000", [50000] * 40), ("96 pairs of 20000", [20000] * 96)]
for name, lens in shapes:
    tasks = []
    for i, n in enumerate(lens):
        a, b, ra, rb = synth_pair(63000 + i, length=n)
        tasks.append(([a, b], [ra, rb], None, None))
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
    steps = max(4, min(48, int(8e11 / cells)))
    line = []
    for env in ({"CSADP_BITS_PACK": "0"}, {"CSADP_BITS_PACK": "1"}):
        best, tm = 0.0, None
        for rep in range(2):
            for k, v in env.items():
                os.environ[k] = v
            csa_amd.reload_config()
            pb = csa_amd.PairBatch(tasks)
            for k in env:
                del os.environ[k]
            csa_amd.reload_config()
            pb.sync()
            for _ in range(2):
                pb.run()
            pb.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                pb.run()
            pb.sync()
            dt = (time.perf_counter() - t0) / steps
            tm = pb.timing()
            best = max(best, cells / dt / 1e12)
            pb.close()
        line.append("%s: %.1f (W%d g%d s%d rec %d)" % (",".join("%s=%s" % (k[11:], v) for k, v in env.items()), best, tm["words_per_lane"], tm["merge_group"], tm["streams"], tm["recoveries"]))
    print("%-40s (%2d steps): TCUPS  %s" % (name, steps, "   ".join(line)), flush=True)
