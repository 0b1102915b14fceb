#!/usr/bin/env python3
"""Whole config 5 (256 pairs, lengths 1 k .. 200 k) as ONE device-resident batch:
time one pass and check the size-independent properties of every result."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import csa_amd  # noqa: E402
from csa_amd.synth import config5_lengths, synth_pair  # noqa: E402
from helpers import degap, rotated, sp_score  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
csa_amd.init(device=0)
la, lb = config5_lengths(256)
tasks = []
for i in range(count):
    a, b, ra, rb = synth_pair(20000 + i, length=min(int(la[i]), cap))
    tasks.append(([a, b], [ra, rb], None, None))
cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
t0 = time.perf_counter()
pb = csa_amd.PairBatch(tasks)
t1 = time.perf_counter()
pb.run()
pb.sync()
t2 = time.perf_counter()
tm = pb.timing()
got = pb.fetch()
t3 = time.perf_counter()
pb.close()
bad = 0
for t, g in zip(tasks, got):
    ok = g["status"] == 0 and degap(g["aligned"][0]) == rotated(t[0][0], t[1][0]) and degap(g["aligned"][1]) == rotated(t[0][1], t[1][1])
    ok = ok and sp_score(g["aligned"]) == g["score"]
    bad += 0 if ok else 1
print("config 5: %d pairs, longest %d, %.3e cells: create %.0f ms, pass %.1f ms (fill %.1f + traceback %.1f) = %.0f GCUPS, fetch %.0f ms, mode %d, %d property failures"
      % (count, max(len(t[0][0]) for t in tasks), cells, (t1 - t0) * 1e3, (t2 - t1) * 1e3, tm["fill_ms"], tm["traceback_ms"],
         cells / (t2 - t1) / 1e9, (t3 - t2) * 1e3, tm["bit_parallel"], bad))
