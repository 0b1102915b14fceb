#!/usr/bin/env python3
"""Host-side cost of the pair API's phases on the bench batch (128 x 16 kbp): create / run+flush /
fetch wall times through the C-ABI, one batch at a time and several in flight."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import config4_tasks  # noqa: E402

csa_amd.init(device=0)
L = csa_amd.lib()
tasks = config4_tasks(0, 128, 16384)
ta = csa_amd.TaskArray(tasks)
res = (csa_amd.Result * ta.n)()
cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
for rnd in range(6):
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    L.csadp_pairs_create(ta.arr, ta.n, ctypes.byref(h))
    t1 = time.perf_counter()
    L.csadp_pairs_run(h)
    L.csadp_pairs_flush(h)
    t2 = time.perf_counter()
    L.csadp_pairs_sync(h)
    t3 = time.perf_counter()
    L.csadp_pairs_fetch(h, res)
    t4 = time.perf_counter()
    for i in range(ta.n):
        L.csadp_free_result(ctypes.byref(res[i]), 2)
    L.csadp_pairs_destroy(h)
    t5 = time.perf_counter()
    print("round %d: create %.3f  run+flush %.3f  sync(device) %.3f  fetch %.3f  free+destroy %.3f ms" % (
        rnd, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3), flush=True)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
for depth in (1, 2, 3, 4, 6):
    print(depth, bench.streaming_leg(csa_amd, tasks, 16, depth=depth), flush=True)
big = config4_tasks(0, 512, 16384)
for depth in (1, 2, 3):
    print("512 pairs", depth, bench.streaming_leg(csa_amd, big, 8, depth=depth), flush=True)
