#!/usr/bin/env python3
"""Strips per workgroup of the bit-parallel fill (CSADP_BITS_CHUNK: auto / 4 / 8 / 16) over batch shapes whose jobs are wider than four strips."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import csa_amd  # noqa: E402
from csa_amd.synth import config5_lengths, synth_pair  # noqa: E402

csa_amd.init(device=0)
shapes = [(120, 17000), (64, 33000), (120, 33000), (40, 50000), (100, 50000), (16, 100000), (64, 100000), (8, 200000), (256, 25000), (512, 9000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for npairs, length in shapes:
    tasks = []
    for i in range(npairs):
        a, b, ra, rb = synth_pair(50000 + i, length=length)
        tasks.append(([a, b], [ra, rb], None, None))
    cells = sum(len(t[0][0]) * len(t[0][1]) for t in tasks)
    steps = max(2, min(12, int(6e11 / cells)))
    line = []
    for env in ({}, {"CSADP_BITS_CHUNK": "4"}, {"CSADP_BITS_CHUNK": "8"}, {"CSADP_BITS_WORDS": "3", "CSADP_BITS_CHUNK": "4"}, {"CSADP_BITS_WORDS": "2", "CSADP_BITS_CHUNK": "4"}):
        for k, v in env.items():
            os.environ[k] = v
        csa_amd.reload_config()
        try:
            pb = csa_amd.PairBatch(tasks)
        except csa_amd.CsadpError:
            pb = None
        for k in env:
            del os.environ[k]
        csa_amd.reload_config()
        if pb is None:
            line.append("%s: -" % (",".join("%s=%s" % (k[11:], v) for k, v in env.items()) or "auto"))
            continue
        pb.sync()
        pb.run()
        pb.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            pb.run()
        pb.sync()
        dt = (time.perf_counter() - t0) / steps
        tm = pb.timing()
        line.append("%s: %.1f (W%d)" % (",".join("%s=%s" % (k[11:], v) for k, v in env.items()) or "auto", cells / dt / 1e12, tm["words_per_lane"]))
        pb.close()
    print("%4d pairs of %6d: TCUPS  %s" % (npairs, length, "   ".join(line)), flush=True)
