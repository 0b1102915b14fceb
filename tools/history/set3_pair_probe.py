#!/usr/bin/env python3
"""Traceback of the reference's Set3 first pair (its two shortest sequences) next to a synthetic pair of the same size."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import csa_amd  # noqa: E402
import helpers as H  # noqa: E402
from csa_amd.synth import synth_pair  # noqa: E402

csa_amd.init(device=0)
_, seqs = H.read_fasta(os.path.join(H.GOLDEN, "data", "Set3.txt"))
rc, rots = csa_amd.find_rotations(seqs)[:2]
order = sorted(range(len(seqs)), key=lambda i: len(seqs[i]))
a, b = order[0], order[1]
tasks = {"Set3 first pair": ([seqs[a], seqs[b]], [rots[a], rots[b]], None, None)}
x, y, ra, rb = synth_pair(777, length=16384)
tasks["synthetic"] = ([x, y], [ra, rb], None, None)
for name, task in tasks.items():
    pb = csa_amd.PairBatch([task])
    best = None
    for _ in range(3):
        pb.run()
        pb.sync()
        t = pb.timing()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    r = pb.fetch()[0]
    pb.close()
    gaps = sum(1 for p, q in zip(r["aligned"][0], r["aligned"][1]) if p == 45 or q == 45)
    print("%s: %d x %d consensus %d gap columns %d: fill %.3f ms traceback+expand %.3f ms" % (
        name, len(task[0][0]), len(task[0][1]), r["consensus"], gaps, best["fill_ms"], best["traceback_ms"]), flush=True)

# the pair csadp_msa's big gap starts with: the two shortest REGIONS of the widest gap of the anchor map
rc, segs, _ = csa_amd.build_anchor_map(seqs, rots)
n = len(seqs)
best = None
for a_, b_ in zip(segs, segs[1:]):
    if not b_[1] and not a_[1]:
        pass
    lens = [b_[2][s] - (a_[2][s] + a_[0]) for s in range(n)]
    if best is None or max(lens) > max(best[0]):
        best = (lens, a_, b_)
lens, sa, sb = best
order = sorted(range(n), key=lambda s: lens[s])
print("widest gap: region lengths", sorted(lens))
def region(s):
    st = sa[2][s] + sa[0]
    t = seqs[s]
    r = rots[s]
    return bytes(t[(r + st + i) % len(t)] for i in range(lens[s]))
p, q = order[0], order[1]
task = ([region(p), region(q)], [0, 0], None, None)
pb = csa_amd.PairBatch([task])
bestt = None
for _ in range(3):
    pb.run(); pb.sync(); t = pb.timing()
    if bestt is None or t["total_ms"] < bestt["total_ms"]:
        bestt = t
r = pb.fetch()[0]
pb.close()
import itertools
ops = ['D' if (x != 45 and y != 45) else ('a' if x == 45 else 'b') for x, y in zip(r["aligned"][0], r["aligned"][1])]
runs = [len(list(g)) for kk, g in itertools.groupby(ops) if kk != 'D']
print("msa's first pair: %d x %d consensus %d gap columns %d in %d runs, longest %s: fill %.3f ms traceback+expand %.3f ms" % (
    lens[p], lens[q], r["consensus"], sum(runs), len(runs), sorted(runs)[-5:], bestt["fill_ms"], bestt["traceback_ms"]), flush=True)
