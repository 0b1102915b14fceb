#!/usr/bin/env python3
"""Timing of csadp_msa on the reference's example sets, second call in the same process (HIP
start-up excluded)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import csa_amd  # noqa: E402
import helpers as H  # noqa: E402

csa_amd.init(device=0)
for name in (sys.argv[1:] or ["Primates", "Mammals", "Set3"]):
    _, seqs = H.read_fasta(os.path.join(H.GOLDEN, "data", name + ".txt"))
    for k in range(3):
        t0 = time.perf_counter()
        rc, rot, rows, st = csa_amd.msa(seqs)
        dt = time.perf_counter() - t0
        print("%s call %d: %.1f ms total  rotations %.1f  anchors %.1f  dp %.1f  rows %.1f  (%d gaps, %d fills, %.2f Gcells)"
              % (name, k, dt * 1e3, st["rotations_ms"], st["anchors_ms"], st["dp_ms"], st["rows_ms"], st["dp_gaps"], st["fills"], st["cells"] / 1e9))
