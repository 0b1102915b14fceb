#!/usr/bin/env python3
"""Cycles per step of ONE matrix as a function of its number of strips (columns / 2048), rows fixed: separates the cost of a lone
wave's step from what the chain of strips adds (hand-offs between waves of a workgroup, between workgroups of a job)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402

csa_amd.init(device=0)
r = random.Random(5)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
a = bytes(r.choice(b"ACGT") for _ in range(rows))
for strips in (1, 2, 4, 5, 8, 12, 16):
    cols = strips * 2048 - 8
    b = bytes(a[:cols])
    pb = csa_amd.PairBatch([([a, b], None, None, None)])
    best = None
    for _ in range(3):
        pb.run()
        pb.sync()
        t = pb.timing()
        if best is None or t["fill_ms"] < best["fill_ms"]:
            best = t
    pb.fetch()
    pb.close()
    steps = rows + 64 + 96 * (strips - 1)
    print("%2d strips x %d rows: fill %.3f ms = %.0f cycles per step (chain of %d steps), traceback %.3f ms"
          % (strips, rows, best["fill_ms"], best["fill_ms"] * 1e-3 * 2.4e9 / steps, steps, best["traceback_ms"]), flush=True)
