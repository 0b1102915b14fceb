#!/usr/bin/env python3
"""Are the host stages of mode N (rotations, anchors) bound by page faults of their fresh multi-megabyte vectors?  Same calls with glibc told to keep
freed memory in the heap (no mmap per large vector, no trim)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import csa_amd
from helpers import GOLDEN, read_fasta
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
if keep:
    libc = ctypes.CDLL("libc.so.6")
    libc.mallopt(-1, 1 << 30)   # M_TRIM_THRESHOLD
    libc.mallopt(-3, 1 << 30)   # M_MMAP_THRESHOLD
for name in ("Primates", "Mammals"):
    _, seqs = read_fasta(os.path.join(GOLDEN, "data", name + ".txt"))
    best = None
    for rep in range(6):
        t0 = time.perf_counter()
        rc, rot, info = csa_amd.find_rotations(seqs)
        t1 = time.perf_counter()
        m = csa_amd.build_anchor_map(seqs, rot)
        t2 = time.perf_counter()
        cur = ((t1 - t0) * 1e3, (t2 - t1) * 1e3)
        if rep >= 2 and (best is None or sum(cur) < sum(best)):
            best = cur
    print("%s (%s): rotations %.2f ms, anchor map %.2f ms" % (name, "heap kept" if keep else "default malloc", best[0], best[1]), flush=True)
