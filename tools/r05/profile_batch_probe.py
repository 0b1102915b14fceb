#!/usr/bin/env python3
"""Throughput of batches of N-SEQUENCE tasks (profile fills i >= 2: nw_fill_cells at full occupancy) through csadp_align_batch:
families of 8 x 4 kbp and of 16 x 16 kbp (round-4 VERDICT item 4).  CSADP_TRACE_HOST=1 prints the rounds' phases."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import csa_amd  # noqa: E402
from helpers import random_family, rng  # noqa: E402

csa_amd.init(device=0)
rotate = "--misrotated" in sys.argv
shapes = [(int(a), int(b), int(c)) for a, b, c in (x.split("x") for x in sys.argv[1:] if not x.startswith("--"))] or [(512, 8, 4000), (64, 16, 16000)]
for nfam, nseq, length in shapes:
    r = rng(nfam * 1000 + nseq)
    t0 = time.perf_counter()
    tasks = []
    for f in range(nfam):
        fam = random_family(r, nseq, length, mut=0.08, indel=0.02)
        tasks.append((fam, [r.randrange(len(x)) for x in fam] if rotate else None, None, None))      # co-linear (what lies between two anchors) unless --misrotated
    gen = time.perf_counter() - t0
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        got = csa_amd.align_batch(tasks)
        dt = time.perf_counter() - t0
        cells = sum(g["cells"] for g in got)
        fills = sum(g["fills"] for g in got)
        assert all(g["status"] == 0 for g in got)
        ph = csa_amd.last_batch_phases()
        print("%d families of %d x %d: call %d: %.1f ms, %.2f Gcells in %d fills = %.1f GCUPS end to end; %d rounds in %d groups: device %.1f ms (= %.0f GCUPS "
              "while the device works), tables %.1f apply %.1f speculate %.1f commit %.1f seed %.1f results %.1f"
              % (nfam, nseq, length, rep, dt * 1e3, cells / 1e9, fills, cells / dt / 1e9, ph["rounds"], ph["round_groups"], ph["device_ms"],
                 cells / max(ph["device_ms"], 1e-9) / 1e6, ph["tables_ms"], ph["apply_ms"], ph["refine_speculate_ms"], ph["refine_commit_ms"],
                 ph["seed_ms"], ph["results_ms"]), "recoveries so far:", csa_amd.recoveries(), flush=True)
