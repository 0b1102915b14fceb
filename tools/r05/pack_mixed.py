#!/usr/bin/env python3
"""Batches of pairs of MIXED widths (one to four strips) with jobs sharing four-wave workgroups (first fit; CSADP_BITS_PACK=1) against one workgroup per job (=0),
and the uniform shapes of tools/r05/pack_shapes.py once more (the table-driven form against the fixed two / four per workgroup measured there)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
r = random.Random(5)
shapes = [("256 pairs of 500-16000", [r.randrange(500, 16000) for _ in range(256)]),
          ("512 pairs of 300-8000", [r.randrange(300, 8000) for _ in range(512)]),
          ("128 pairs of 2000-16384", [r.randrange(2000, 16384) for _ in range(128)]),
          ("200 pairs: 150 of 12000, 50 of 3000", [12000] * 150 + [3000] * 50),
          ("256 pairs of 8000", [8000] * 256), ("64 pairs of 8000", [8000] * 64), ("128 pairs of 12000", [12000] * 128), ("512 pairs of 5000", [5000] * 512),
          ("128 pairs of 16384", [16384] * 128)]
for name, lens in shapes:
    tasks = []
    for i, n in enumerate(lens):
        a, b, ra, rb = synth_pair(61000 + i, length=n)
        tasks.append(([a, b], [ra, rb], None, None))
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
    steps = max(8, min(64, int(8e11 / cells)))
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
            for _ in range(4):
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
        line.append("%s: %.1f (W%d g%d s%d)" % (",".join("%s=%s" % (k[11:], v) for k, v in env.items()), best, tm["words_per_lane"], tm["merge_group"], tm["streams"]))
    print("%-40s (%2d steps): TCUPS  %s" % (name, steps, "   ".join(line)), flush=True)
