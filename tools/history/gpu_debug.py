"""First-contact GPU check: runs growing cases and prints the first mismatches in detail."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import csa_amd
from helpers import *

csa_amd.init(device=0)
print(csa_amd.device_info(), flush=True)
bad = 0
cases = load_golden("tiny_pairs.json")
t0 = time.time()
got = csa_amd.align_batch([golden_task(c) for c in cases])
print("tiny pairs batch: %.3fs" % (time.time() - t0), flush=True)
for c, g in zip(cases, got):
    exp = golden_aligned(c)
    if g["aligned"] != (exp if exp[0] is not None else None) or g["status"] != 0:
        bad += 1
        if bad <= 5:
            print("MISMATCH pair", c["texts"], c["rots"], c["starts"], c["ends"], "\n  got", g, "\n  exp", exp, c["consensus"], flush=True)
print("tiny pairs bad:", bad, "of", len(cases), flush=True)
bad = 0
cases = load_golden("tiny_families.json")
got = csa_amd.align_batch([golden_task(c) for c in cases])
for c, g in zip(cases, got):
    exp = golden_aligned(c)
    if g["aligned"] != (exp if exp[0] is not None else None) or g["status"] != 0:
        bad += 1
        if bad <= 3:
            print("MISMATCH fam", c["texts"], c["rots"], c["starts"], c["ends"], "\n  got", g, "\n  exp", exp, flush=True)
print("tiny families bad:", bad, "of", len(cases), flush=True)
r = rng(7)
for length in [100, 500, 1023, 1024, 1025, 2000, 4096, 8000]:
    fam = random_family(r, 2, length, mut=0.1, indel=0.04)
    rots = [r.randrange(len(f)) for f in fam]
    t0 = time.time()
    g = csa_amd.align_batch([(fam, rots, None, None)])[0]
    dt = time.time() - t0
    cons, strs, st = oracle_progressive(fam, rots)
    ok = (g["aligned"] == strs and g["score"] == st.last_score)
    print("len", length, [len(f) for f in fam], "ok" if ok else "MISMATCH", g["score"], st.last_score, g["consensus"], cons, "%.3fs" % dt, flush=True)
    if not ok and g["aligned"]:
        for i in range(2):
            x, y = g["aligned"][i], strs[i]
            k = next((j for j in range(min(len(x), len(y))) if x[j] != y[j]), None)
            print("   first diff seq", i, "at", k, len(x), len(y))
a, b, ra, rb = synth_pair(1)
pb = csa_amd.PairBatch([([a, b], [ra, rb], None, None)])
for it in range(3):
    pb.run(); pb.sync(); print("16k pair timing", pb.timing(), flush=True)
g = pb.fetch()[0]
print("16k pair", g["score"], g["consensus"], sp_score(g["aligned"]), degap(g["aligned"][0]) == rotated(a, ra), degap(g["aligned"][1]) == rotated(b, rb), flush=True)
