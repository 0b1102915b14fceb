#!/usr/bin/env python3
"""Config 5 (256 pairs of 1-200 kbp) with the partial last chunks and narrow jobs sharing workgroups (CSADP_BITS_PACK=1) against one workgroup per chunk (=0), alternating."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import config5_lengths, synth_pair  # noqa: E402

csa_amd.init(device=0)
la, lb = config5_lengths()
tasks = []
for p, n in enumerate(la):
    a, b, ra, rb = synth_pair(50000 + p, length=n)
    tasks.append(([a, b], [ra, rb], None, None))
cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for pack in ("0", "1"):
        os.environ["CSADP_BITS_PACK"] = pack
        csa_amd.reload_config()
        pb = csa_amd.PairBatch(tasks)
        pb.sync()
        pb.run()
        pb.sync()
        t0 = time.perf_counter()
        for _ in range(4):
            pb.run()
        pb.sync()
        dt = (time.perf_counter() - t0) / 4
        tm = pb.timing()
        pb.close()
        print("PACK=%s: %.1f TCUPS  %.2f ms/step (W%d g%d s%d, %d fill workgroups' entries, recoveries %d)" % (pack, cells / dt / 1e12, dt * 1e3, tm["words_per_lane"], tm["merge_group"], tm["streams"], tm.get("fill_tiles", 0), tm["recoveries"]), flush=True)
