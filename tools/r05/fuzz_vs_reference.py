#!/usr/bin/env python3
"""Differential fuzz on the GPU box: random ProgressiveDP tasks through csadp_align_batch against the COMPILED REFERENCE
(oracle/_ref/libcsa_ref.so, the unmodified dynamicprogramming.c): strings, consensus and progress tokens must be identical.
Shapes the fixed suites only sample: 2..24 sequences, regions of 0..4000 letters, equal lengths (stale borders, quirk Q1), empty
regions, sub-regions with rotations, homopolymers and short periods (ties everywhere), one long sequence among short ones.

A third argument "pairs" fuzzes the device-I/O pair path instead (nw_pack_planes, nw_fill_bits at 1-4 words per lane and 4 / 8 / 16 strips per
workgroup, nw_traceback_windows, nw_expand_rows): batches of 2-sequence tasks only, 1..9000 letters, sub-regions and rotations, run twice per batch.

usage: python tools/r05/fuzz_vs_reference.py [seconds] [seed] [pairs|long]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import csa_amd  # noqa: E402
from helpers import have_ref, random_family, ref_progressive, rng  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
assert have_ref(), "oracle/_ref/libcsa_ref.so did not travel"
csa_amd.init(device=0)
r = rng(seed)


def make_task():
    kind = r.choice(["family", "family", "family", "pair", "equal", "periodic", "skewed", "subregions", "tiny"])
    if LONG and r.random() < 0.04:
        kind = "long"
    if kind == "pair":
        fam = random_family(r, 2, r.choice([1, 5, 60, 300, 1500, 4000]), mut=r.choice([0.0, 0.05, 0.3, 0.75]), indel=r.choice([0.0, 0.05, 0.3]))
    elif kind == "equal":
        n = r.choice([3, 4, 6, 9])
        fam = random_family(r, n, r.choice([30, 200, 900]), mut=r.choice([0.05, 0.2]), indel=r.choice([0.0, 0.05]))
        m = min(len(f) for f in fam) or 1
        fam = [f[:m] if f else b"A" for f in fam]
    elif kind == "periodic":
        n = r.choice([2, 3, 5])
        unit = bytes(r.choice(b"ACGT") for _ in range(r.choice([1, 2, 3, 7])))
        fam = [unit * r.randrange(1, 200 // len(unit) + 2) for _ in range(n)]
    elif kind == "skewed":
        n = r.choice([3, 5, 8])
        fam = random_family(r, n, 40, mut=0.2, indel=0.1)
        fam[r.randrange(n)] = bytes(r.choice(b"ACGT") for _ in range(r.choice([800, 2500])))
    elif kind == "long":                                 # matrices of 40-100 strips: chains of 10-25 workgroups of nw_fill_cells
        fam = random_family(r, r.choice([3, 4, 5]), r.choice([5000, 8000, 12000]), mut=r.choice([0.03, 0.15]), indel=r.choice([0.01, 0.05]))
    elif kind == "tiny":
        n = r.choice([2, 3, 4, 12, 24])
        fam = [bytes(r.choice(b"ACGT") for _ in range(r.randrange(1, 6))) for _ in range(n)]
    else:
        n = r.choice([3, 3, 4, 5, 8, 13, 24]) if kind == "family" else r.choice([3, 5])
        fam = random_family(r, n, r.choice([8, 50, 250, 1000, 2500] if n <= 8 else [8, 50, 250]), mut=r.choice([0.02, 0.1, 0.3]),
                            indel=r.choice([0.0, 0.03, 0.15, 0.4]))
    fam = [f if f else b"G" for f in fam]
    rots = [r.randrange(len(f)) for f in fam]
    starts, ends = [0] * len(fam), [len(f) for f in fam]
    if kind == "subregions" or r.random() < 0.15:
        for i, f in enumerate(fam):
            if r.random() < 0.6:
                a = r.randrange(len(f) + 1)
                starts[i], ends[i] = a, r.randrange(a, len(f) + 1)
    return (fam, rots, starts, ends)


def make_pair():
    base = r.choice([1, 3, 40, 700, 2100, 4200, 6300, 9000])
    fam = random_family(r, 2, base, mut=r.choice([0.0, 0.03, 0.1, 0.4, 0.75]), indel=r.choice([0.0, 0.01, 0.05, 0.3]))
    if r.random() < 0.1:
        fam[1] = bytes(r.choice(b"ACGT") for _ in range(r.randrange(1, 3000)))       # unrelated partner of another length
    fam = [f if f else b"T" for f in fam]
    rots = [r.randrange(len(f)) for f in fam]
    starts, ends = [0, 0], [len(f) for f in fam]
    if r.random() < 0.2:
        for i, f in enumerate(fam):
            a = r.randrange(len(f) + 1)
            starts[i], ends[i] = a, r.randrange(a, len(f) + 1)
    return (fam, rots, starts, ends)


pairs_mode = len(sys.argv) > 3 and sys.argv[3] == "pairs"
LONG = len(sys.argv) > 3 and sys.argv[3] == "long"      # families mode with a few long families per batch
t_end = time.time() + budget
done = bad = 0
cells = 0
while time.time() < t_end:
    if pairs_mode:
        tasks = [make_pair() for _ in range(r.choice([1, 7, 64, 200]))]
        env = {"CSADP_BITS_WORDS": r.choice(["1", "2", "3", "4", None]), "CSADP_BITS_CHUNK": r.choice(["4", "8", "16", None])}
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
        csa_amd.reload_config()
        pb = csa_amd.PairBatch(tasks)
        pb.run()
        pb.run()
        got = pb.fetch()
        pb.close()
        for k in env:
            os.environ.pop(k, None)
        csa_amd.reload_config()
    else:
        tasks = [make_task() for _ in range(r.choice([40, 120, 160, 300]))]      # 128 and more: four round groups
        order = r.choice(["0", "1", None])                                       # work list of nw_fill_cells: job by job, level by level, by the rule
        if order is not None:
            os.environ["CSADP_CELLS_ORDER"] = order
        csa_amd.reload_config()
        got = csa_amd.align_batch(tasks)
        os.environ.pop("CSADP_CELLS_ORDER", None)
        csa_amd.reload_config()
    for t, g in zip(tasks, got):
        cons, strs, _ = ref_progressive(*t)
        same = g["status"] == 0 and ((g["aligned"] is None and all(s is None for s in strs)) or g["aligned"] == strs)
        if same and g["aligned"] is not None:
            same = g["consensus"] == cons
        if not same:
            bad += 1
            print("MISMATCH", [x.decode() for x in t[0]], t[1], t[2], t[3], g["status"], flush=True)
        done += 1
        cells += g.get("cells", 0)
    print("%d tasks, %d mismatches, %.2f Gcells, %.0f s left" % (done, bad, cells / 1e9, t_end - time.time()), flush=True)
print("fuzz against the compiled reference: %d tasks, %d mismatches (seed %d)" % (done, bad, seed))
sys.exit(1 if bad else 0)
