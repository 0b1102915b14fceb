#!/usr/bin/env python3
"""Words per lane with jobs of one or two strips packed into four-wave workgroups: auto (the rule) against 1 / 2 / 3 / 4 words forced."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(200, 7000), (96, 8192), (128, 4000), (300, 2000), (2048, 1500), (64, 8000), (256, 8000), (512, 5000), (1024, 3000), (128, 12000), (256, 12000), (128, 16384), (64, 16384), (32, 16384)]
for npairs, length in shapes:
    tasks = []
    for i in range(npairs):
        a, b, ra, rb = synth_pair(61000 + i, length=length)
        tasks.append(([a, b], [ra, rb], None, None))
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
    steps = max(8, min(64, int(8e11 / cells)))
    line = []
    for env in ({}, {"CSADP_BITS_WORDS": "1"}, {"CSADP_BITS_WORDS": "2"}, {"CSADP_BITS_WORDS": "3"}, {"CSADP_BITS_WORDS": "4"}):
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
        line.append("%s: %.1f (W%d g%d s%d)" % (",".join("%s=%s" % (k[11:], v) for k, v in env.items()) or "auto", best, tm["words_per_lane"], tm["merge_group"], tm["streams"]))
    print("%4d pairs of %6d (%2d steps): TCUPS  %s" % (npairs, length, steps, "   ".join(line)), flush=True)
