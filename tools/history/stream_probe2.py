#!/usr/bin/env python3
"""Enqueue N one-pass pair batches back to back (create + run + flush), then fetch them all:
does the device overlap batches that are all queued up front?"""
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
for N in (1, 2, 3, 4, 4, 3, 2, 1):
    hs = []
    t0 = time.perf_counter()
    for _ in range(N):
        h = ctypes.c_void_p()
        L.csadp_pairs_create(ta.arr, ta.n, ctypes.byref(h))
        L.csadp_pairs_run(h)
        L.csadp_pairs_flush(h)
        hs.append(h)
    t1 = time.perf_counter()
    for h in hs:
        L.csadp_pairs_sync(h)
    t2 = time.perf_counter()
    for h in hs:
        L.csadp_pairs_fetch(h, res)
        for i in range(ta.n):
            L.csadp_free_result(ctypes.byref(res[i]), 2)
        L.csadp_pairs_destroy(h)
    t3 = time.perf_counter()
    print("N=%d: enqueue %.2f ms, wait %.2f ms (%.2f per batch), fetch+destroy %.2f ms" % (
        N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3 / N, (t3 - t2) * 1e3), flush=True)
