#!/usr/bin/env python3
"""Where should the fetcher layout of nw_fill_cells end?  Batches of N matrices of 5000 x 6187 (13 workgroups each) filled with the layout forced on
(CSADP_CELLS_FETCH=100000) and off (0): one workgroup of six waves per compute unit, the rest queueing, against two workgroups of four waves per unit."""
import os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ["CSADP_BITS"] = "0"
    os.environ["CSADP_PK16"] = "0"
    import csa_amd
    csa_amd.init(device=0)
    rnd = random.Random(7)
    seq = lambda n: bytes(rnd.choice(b"ACGT") for _ in range(n))
    for njobs in (8, 16, 19, 24, 32, 48, 96):
        tasks = [([seq(6187), seq(5000)], [0, 0], None, None) for _ in range(njobs)]
        pb = csa_amd.PairBatch(tasks)
        best = None
        for _ in range(3):
            pb.run(); pb.sync()
            t = pb.timing()
            best = t if best is None or t["fill_ms"] < best["fill_ms"] else best
        pb.close()
        print("%3d jobs (%4d workgroups): fill %.3f ms, traceback %.3f ms" % (njobs, njobs * 13, best["fill_ms"], best["traceback_ms"]), flush=True)
else:
    for fetch in ("0", "100000"):
        print("CSADP_CELLS_FETCH=%s" % fetch, flush=True)
        env = dict(os.environ, CSADP_CELLS_FETCH=fetch)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
