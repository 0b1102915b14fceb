#!/usr/bin/env python3
"""Where a strip of nw_fill_cells waits: a library built with -DCSADP_CELL_TIMERS (tools/r04/cells_times.sh) records, at every strip's
middle block, the 100 MHz time at entry, with the hand-off in hand and at the block's end.  Prints the lag of each strip behind its left
neighbour in time, split by kind of hand-off (inside a workgroup / between workgroups), the wait and the block's own duration."""
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("CSADP_BITS", "0")
os.environ.setdefault("CSADP_PK16", "0")
import csa_amd  # noqa: E402

csa_amd.init(device=0)
lib = ctypes.CDLL(csa_amd.LIB_PATH)
rnd = random.Random(5)


def seq(n):
    return bytes(rnd.choice(b"ACGT") for _ in range(n))


for nrows, ncols in [(16384, 16384), (16979, 20852), (5000, 6187)]:
    task = ([seq(ncols), seq(nrows)], [0, 0], None, None)
    pb = csa_amd.PairBatch([task])
    for _ in range(3):
        pb.run()
        pb.sync()
    t = pb.timing()
    strips = (ncols + 127) // 128
    buf = (ctypes.c_ulonglong * (12 * strips))()
    rc = lib.csadp_debug_cell_times(buf, strips)
    assert rc == 0, rc
    pb.close()
    rows = [buf[12 * s:12 * s + 12] for s in range(strips)]
    inwg, cross, waits_in, waits_x, durs, xsame = [], [], [], [], [], []
    for s in range(1, strips):
        lag = (rows[s][1] - rows[s - 1][1]) * 10e-3        # us between "hand-off in hand" of neighbours at the same block
        wait = (rows[s][4] - rows[s][3])                   # shader clocks spent waiting at this block
        dur = (rows[s][5] - rows[s][4])
        durs.append(dur)
        if s % 4 == 0:
            cross.append(lag)
            waits_x.append(wait)
            xsame.append(rows[s][6] == rows[s - 1][6])
        else:
            inwg.append(lag)
            waits_in.append(wait)
    med = lambda v: sorted(v)[len(v) // 2] if v else 0
    print("%d x %d (%d strips): fill %.3f ms; lag inside a workgroup median %.2f us (min %.2f max %.2f), between workgroups median %.2f us (min %.2f max %.2f; %d of %d pairs on one XCC)"
          % (nrows, ncols, strips, t["fill_ms"], med(inwg), min(inwg), max(inwg), med(cross), min(cross), max(cross), sum(xsame), len(xsame)))
    print("    wait at the block: inside median %d clocks, between workgroups median %d; the block itself median %d clocks (100 MHz ticks per block: %d)"
          % (med(waits_in), med(waits_x), med(durs), med([r[2] - r[1] for r in rows])))
    per = lambda ss: med([(rows[s][7] - rows[s][5]) / 16.0 for s in ss])
    own = lambda ss: med([rows[s][5] - rows[s][4] for s in ss])
    wt = lambda ss: med([rows[s][4] - rows[s][3] for s in ss])
    kinds = {"first strip": [0], "first of a chunk": list(range(4, strips, 4)), "inside (ring)": [s for s in range(1, strips) if s % 4]}
    stmt = lambda ss: med([rows[s][11] for s in ss])
    for k, ss in kinds.items():
        print("    %-18s period %.0f clocks per block, of them the block itself %d (the assembly statement %d), the wait %d" % (k, per(ss), own(ss), stmt(ss), wt(ss)))
    print("    lag behind the left strip in us, strip by strip: " + " ".join("%.1f" % ((rows[s][1] - rows[s - 1][1]) * 10e-3) for s in range(1, min(strips, 41))))
    us = lambda a, b: (a - b) * 10e-3
    t0 = rows[0][8]
    last = strips - 1
    print("    the whole fill %.1f us: the last strip enters %.1f us after the first, passes block 2 at +%.1f, runs %.1f us (the first strip %.1f us)"
          % (us(rows[last][9], t0), us(rows[last][8], t0), us(rows[last][10], rows[last][8]), us(rows[last][9], rows[last][8]), us(rows[0][9], t0)))
    print("    lag at block 2 (end), strip by strip: " + " ".join("%.1f" % us(rows[s][10], rows[s - 1][10]) for s in range(1, min(strips, 41))))
    print("    lag at the last block's end:          " + " ".join("%.1f" % us(rows[s][9], rows[s - 1][9]) for s in range(1, min(strips, 41))))
    durs_all = [us(rows[s][9], rows[s][10]) for s in range(strips)]
    print("    blocks 3.. of a strip take (us): first %.1f, median %.1f, max %.1f at strip %d" % (durs_all[0], med(durs_all), max(durs_all), durs_all.index(max(durs_all))))
    same = [l for l, x in zip(cross, xsame) if x]
    diff = [l for l, x in zip(cross, xsame) if not x]
    if same and diff:
        print("    between workgroups on one XCC: median %.2f us; on two: %.2f us" % (med(same), med(diff)))
