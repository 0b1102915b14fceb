#!/usr/bin/env python3
"""Config 5 as one batch under forced shapes: words per lane x strips per workgroup of the chunked fill."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import config5_lengths, synth_pair  # noqa: E402

csa_amd.init(device=0)
la, _ = config5_lengths(256)
tasks = []
for i, length in enumerate(la):
    a, b, ra, rb = synth_pair(20000 + i, length=int(length))
    tasks.append(([a, b], [ra, rb], None, None))
cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
combos = [{}] + [{"CSADP_BITS_WORDS": w, "CSADP_BITS_CHUNK": c} for w in ("1", "2", "3") for c in ("4", "8", "16")]
combos += [dict(x, CSADP_BITS_STREAMS="2", CSADP_BITS_GROUP="1") for x in ({"CSADP_BITS_WORDS": "2", "CSADP_BITS_CHUNK": "4"}, {"CSADP_BITS_WORDS": "3", "CSADP_BITS_CHUNK": "4"})]
for rep in range(2):
    for env in combos:
        for k, v in env.items():
            os.environ[k] = v
        csa_amd.reload_config()
        try:
            pb = csa_amd.PairBatch(tasks)
        except csa_amd.CsadpError as e:
            print("%-70s: %s" % (env, e), flush=True)
            pb = None
        for k in env:
            del os.environ[k]
        csa_amd.reload_config()
        if pb is None:
            continue
        pb.sync()
        pb.run()
        pb.sync()
        t0 = time.perf_counter()
        for _ in range(4):
            pb.run()
        pb.sync()
        dt = (time.perf_counter() - t0) / 4
        tm = pb.timing()
        print("%-70s: %.2f ms per pass = %.1f TCUPS (W %d, passes/launch %d, streams %d)" % (env, dt * 1e3, cells / dt / 1e12, tm["words_per_lane"], tm["merge_group"], tm["streams"]), flush=True)
        pb.close()
