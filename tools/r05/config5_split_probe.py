#!/usr/bin/env python3
"""Config 5 (256 pairs of 1-200 kbp) as ONE batch against the same pairs as TWO batches in flight together: the long jobs (whose chain of
strips bounds a pass) in a shape of their own, the rest in theirs.  Environment switches are re-read between the two creates."""
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
order = sorted(range(256), key=lambda i: -la[i])


def make(ids, env):
    for k, v in env.items():
        os.environ[k] = v
    csa_amd.reload_config()
    pb = csa_amd.PairBatch([tasks[i] for i in ids])
    for k in env:
        del os.environ[k]
    csa_amd.reload_config()
    return pb


def timed(batches, steps=3, warmup=1):
    for pb in batches:
        pb.sync()
    for _ in range(warmup):
        for pb in batches:
            pb.run()
    for pb in batches:
        pb.flush()
    for pb in batches:
        pb.sync()
    t0 = time.perf_counter()
    for pb in batches:
        for _ in range(steps):
            pb.run()
    for pb in batches:
        pb.flush()
    for pb in batches:
        pb.sync()
    dt = (time.perf_counter() - t0) / steps
    tms = [pb.timing() for pb in batches]
    return dt, tms


SHAPE = {"CSADP_BITS_STREAMS": "2", "CSADP_BITS_GROUP": "1"}       # four slots per batch: two batches of this size must fit the HBM together
one = make(list(range(256)), {})
dt, tms = timed([one])
print("one batch (the engine's own shape): %.2f ms per pass = %.1f TCUPS (W %d, passes/launch %d)" % (dt * 1e3, cells / dt / 1e12, tms[0]["words_per_lane"], tms[0]["merge_group"]), flush=True)
ref = one.fetch()
one.close()
one = make(list(range(256)), SHAPE)
dt, tms = timed([one])
print("one batch, 2 streams x 1 pass: %.2f ms per pass = %.1f TCUPS (W %d)" % (dt * 1e3, cells / dt / 1e12, tms[0]["words_per_lane"]), flush=True)
one.close()
for nlong in (8, 16, 32, 48, 64):
    for envl in ({}, {"CSADP_BITS_WORDS": "1"}, {"CSADP_BITS_WORDS": "2"}, {"CSADP_BITS_WORDS": "1", "CSADP_BITS_CHUNK": "4"}, {"CSADP_BITS_WORDS": "2", "CSADP_BITS_CHUNK": "4"}):
        L = make(order[:nlong], dict(SHAPE, **envl))
        S = make(order[nlong:], SHAPE)
        dt, tms = timed([L, S])
        print("%2d longest apart %-50s: %.2f ms per pass = %.1f TCUPS  (long: W %d fill %.2f ms; rest: W %d fill %.2f ms)" % (
            nlong, envl, dt * 1e3, cells / dt / 1e12, tms[0]["words_per_lane"], tms[0]["fill_ms"], tms[1]["words_per_lane"], tms[1]["fill_ms"]), flush=True)
        if nlong == 16 and not envl:
            gl, gs = L.fetch(), S.fetch()
            ok = all(gl[k]["aligned"] == ref[i]["aligned"] for k, i in enumerate(order[:nlong])) and all(gs[k]["aligned"] == ref[i]["aligned"] for k, i in enumerate(order[nlong:]))
            print("   results equal to the one-batch run:", ok, flush=True)
        L.close()
        S.close()
