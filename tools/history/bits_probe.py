#!/usr/bin/env python3
"""Timing probe for pair batches (no verification): tools/bits_probe.py [--lib path] [--pairs N] [--len L] [--passes K]
Used for ablation builds of the kernels (e.g. direction stores compiled out)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csa_amd  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib")
ap.add_argument("--pairs", type=int, default=128)
ap.add_argument("--len", type=int, default=16384, dest="length")
ap.add_argument("--passes", type=int, default=20)
a = ap.parse_args()
if a.lib:
    csa_amd.LIB_PATH = os.path.abspath(a.lib)
csa_amd.init(device=0)
tasks = []
for p in range(a.pairs):
    x, y, ra, rb = synth_pair(p, a.length)
    tasks.append(([x, y], [ra, rb], None, None))
pb = csa_amd.PairBatch(tasks)
for _ in range(4):
    pb.run()
pb.sync()
t0 = time.perf_counter()
for _ in range(a.passes):
    pb.run()
pb.sync()
dt = (time.perf_counter() - t0) / a.passes
pb.run()
pb.sync()
tm = pb.timing()
cells = tm["cells"]
print("lib=%s pairs=%d len=%d: %.3f ms/pass pipelined = %.1f GCUPS; alone fill %.3f ms tb %.3f ms = %.1f GCUPS"
      % (a.lib or "default", a.pairs, a.length, dt * 1e3, cells / dt / 1e9, tm["fill_ms"], tm["traceback_ms"],
         cells / (tm["total_ms"] * 1e-3) / 1e9))
