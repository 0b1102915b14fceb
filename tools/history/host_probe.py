#!/usr/bin/env python3
"""Host-boundary timing of one pair batch: create (validate, pack, upload), one pass, fetch
(download, strings).  CSADP_TRACE_HOST=1 prints the library's own phase timers."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
tasks = []
for p in range(128):
    x, y, ra, rb = synth_pair(p, 16384)
    tasks.append(([x, y], [ra, rb], None, None))
for k in range(3):
    t0 = time.perf_counter()
    pb = csa_amd.PairBatch(tasks)
    t1 = time.perf_counter()
    pb.run()
    pb.sync()
    t2 = time.perf_counter()
    res = pb.fetch()
    t3 = time.perf_counter()
    pb.close()
    print("create %.2f ms  pass %.2f ms  fetch %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
