#!/usr/bin/env python3
"""Latency of ONE matrix (config 2: a 16 kbp pair; config 5's upper end: 100 / 200 kbp pairs): device
fill + traceback of a one-job batch, and the host-to-host time of csadp_align_batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
for length in [int(x) for x in (sys.argv[1:] or ["16384", "100000", "200000"])]:
    a, b, ra, rb = synth_pair(777, length=length)
    task = ([a, b], [ra, rb], None, None)
    pb = csa_amd.PairBatch([task])
    best = None
    for _ in range(3):
        pb.run()
        pb.sync()
        t = pb.timing()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    pb.fetch()
    pb.close()
    t0 = time.perf_counter()
    csa_amd.align_batch([task])
    t1 = time.perf_counter()
    t2 = time.perf_counter()
    csa_amd.align_batch([task])
    t3 = time.perf_counter()
    cells = len(a) * len(b)
    print("%7d x %7d (W %d): fill %.3f ms  traceback+expand %.3f ms  = %.1f GCUPS on the device; align_batch host-to-host %.2f ms (second call %.2f)"
          % (len(a), len(b), best["words_per_lane"], best["fill_ms"], best["traceback_ms"], cells / best["total_ms"] / 1e6, (t1 - t0) * 1e3, (t3 - t2) * 1e3), flush=True)
