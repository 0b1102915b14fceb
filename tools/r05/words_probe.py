#!/usr/bin/env python3
"""Words per lane (CSADP_BITS_WORDS auto / 1 / 2 / 3) over batch shapes, with the round-5 rule of four strips per workgroup."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
shapes = [(2, 200000), (8, 200000), (4, 100000), (16, 100000), (64, 100000), (8, 50000), (40, 50000), (16, 33000), (64, 33000), (200, 33000), (32, 17000), (66, 17000), (120, 17000), (300, 17000), (1000, 5000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for npairs, length in shapes:
    tasks = []
    for i in range(npairs):
        a, b, ra, rb = synth_pair(60000 + i, length=length)
        tasks.append(([a, b], [ra, rb], None, None))
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
    steps = max(3, min(16, int(8e11 / cells)))
    line = []
    for env in ({}, {"CSADP_BITS_WORDS": "1"}, {"CSADP_BITS_WORDS": "2"}, {"CSADP_BITS_WORDS": "3"}):
        best = 0.0
        w = 0
        for rep in range(2):
            for k, v in env.items():
                os.environ[k] = v
            csa_amd.reload_config()
            try:
                pb = csa_amd.PairBatch(tasks)
            except csa_amd.CsadpError:
                pb = None
            for k in env:
                del os.environ[k]
            csa_amd.reload_config()
            if pb is None:
                break
            pb.sync()
            pb.run()
            pb.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                pb.run()
            pb.sync()
            dt = (time.perf_counter() - t0) / steps
            w = pb.timing()["words_per_lane"]
            best = max(best, cells / dt / 1e12)
            pb.close()
        line.append("%s: %.1f%s" % (env.get("CSADP_BITS_WORDS", "auto"), best, (" (W%d)" % w) if not env else ""))
    print("%4d pairs of %6d (%2d steps): TCUPS  %s" % (npairs, length, steps, "   ".join(line)), flush=True)
