#!/usr/bin/env python3
"""Jobs of one or two strips sharing four-wave workgroups of nw_fill_bits (CSADP_BITS_PACK): batches of small pairs, unpacked against packed in a few launch shapes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(256, 8000), (128, 8000), (64, 8000), (512, 5000), (200, 7000), (1024, 3000), (2048, 1500), (128, 4000), (96, 8192), (300, 2000)]
for npairs, length in shapes:
    tasks = []
    for i in range(npairs):
        a, b, ra, rb = synth_pair(61000 + i, length=length)
        tasks.append(([a, b], [ra, rb], None, None))
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
    steps = max(8, min(64, int(8e11 / cells)))
    line = []
    for env in ({"CSADP_BITS_PACK": "0"}, {"CSADP_BITS_PACK": "1"}, {"CSADP_BITS_PACK": "1", "CSADP_BITS_GROUP": "4"}, {"CSADP_BITS_PACK": "1", "CSADP_BITS_GROUP": "8"},
                {"CSADP_BITS_PACK": "1", "CSADP_BITS_STREAMS": "3"}, {"CSADP_BITS_PACK": "1", "CSADP_BITS_STREAMS": "4"}):
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
    print("%4d pairs of %6d (%2d steps): TCUPS  %s" % (npairs, length, steps, "   ".join(line)), flush=True)
