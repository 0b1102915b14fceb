#!/usr/bin/env python3
"""Step cost of the cell-per-lane kernel (nw_fill_cells) from matrix shapes that isolate it:
one strip x many rows (pure step), one workgroup, several chunks, many strips x few rows (strip lag).
Run with CSADP_BITS=0 CSADP_PK16=0 so that pairs take the general kernel."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CSADP_BITS", "0")
os.environ.setdefault("CSADP_PK16", "0")
import csa_amd  # noqa: E402

csa_amd.init(device=0)
rnd = random.Random(5)


def seq(n):
    return bytes(rnd.choice(b"ACGT") for _ in range(n))


SHAPES = [(32768, 128), (32768, 512), (32768, 1024), (32768, 2048), (64, 16384), (64, 32768), (5000, 6187), (16384, 16384)]
for nrows, ncols in SHAPES:
    # the column sequence is the first of the pair, the row sequence the second
    task = ([seq(ncols), seq(nrows)], [0, 0], None, None)
    pb = csa_amd.PairBatch([task])
    best = None
    for _ in range(3):
        pb.run()
        pb.sync()
        t = pb.timing()
        if best is None or t["fill_ms"] < best["fill_ms"]:
            best = t
    if os.environ.get('CSADP_CELL_STATS'): pb.fetch()
    pb.close()
    strips = (ncols + 127) // 128
    steps96 = strips * 96 + nrows
    steps64 = strips * 64 + nrows
    cyc = best["fill_ms"] * 2.4e6
    print("%6d rows x %6d cols (%4d strips): fill %.3f ms = %.0f cycles per step at lag 96 (%d steps), %.0f at lag 64; traceback %.3f ms"
          % (nrows, ncols, strips, best["fill_ms"], cyc / steps96, steps96, cyc / steps64, best["traceback_ms"]), flush=True)
