#!/usr/bin/env python3
"""How quickly do traceback paths of the profile steps coalesce?  (design probe for the band-parallel traceback, CPU only)
For every fill of a progressive task the oracle's direction matrix is walked from sparse start columns of every band of
R rows; a band is 'merged' when the two starts that flank the TRUE path's entry column leave the band in the same column."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import csa_amd  # noqa: E402
import helpers as H  # noqa: E402

RS = [(64, 32), (64, 64), (128, 32), (128, 64), (128, 128), (256, 64), (256, 128)]
report = []


def band_exits(D, nrows, ncols, R, S):
    nb = (nrows + R - 1) // R
    starts = np.unique(np.concatenate([np.arange(0, ncols + 1, S), [ncols]]))
    J = np.repeat(np.minimum((np.arange(nb) + 1) * R, nrows), len(starts)).astype(np.int64)
    top = np.repeat(np.arange(nb) * R, len(starts)).astype(np.int64)
    K = np.tile(starts, nb).astype(np.int64)
    live = (J > top) & (K > 0)
    steps = 0
    while live.any():
        idx = np.nonzero(live)[0]
        d = D[J[idx], K[idx]]
        J[idx] -= (d != 76)
        K[idx] -= (d != 85)
        live[idx] = (J[idx] > top[idx]) & (K[idx] > 0)
        steps += 1
    return starts, K.reshape(nb, len(starts)), J.reshape(nb, len(starts)), steps


def make_filler():
    lib = H.oracle_lib()

    def fill(user, nrows, ncols, nprev, sv, rowcodes, top, left_i, ops, nops, remj, remk, score):
        Hm = (ctypes.c_int * ((nrows + 1) * (ncols + 1)))()
        Dm = ctypes.create_string_buffer((nrows + 1) * (ncols + 1))
        rc = lib.odp_fill(nrows, ncols, rowcodes, sv, nprev, top, left_i, Hm, Dm)
        if rc != 0:
            return -5
        D = np.frombuffer(Dm, dtype=np.uint8, count=(nrows + 1) * (ncols + 1)).reshape(nrows + 1, ncols + 1)
        j, k, n = nrows, ncols, 0
        path = {}
        while j > 0 and k > 0:
            path.setdefault(j, k)            # first (rightmost) column seen in row j = entry column of the row
            d = D[j, k]
            if d == 68:
                ops[n] = 2; j -= 1; k -= 1
            elif d == 76:
                ops[n] = 1; k -= 1
            else:
                ops[n] = 0; j -= 1
            n += 1
        nops[0] = n; remj[0] = j; remk[0] = k
        score[0] = Hm[nrows * (ncols + 1) + ncols]
        line = "fill %5d x %5d nprev %2d ops %5d:" % (nrows, ncols, nprev, n)
        for R, S in RS:
            starts, EK, EJ, steps = band_exits(D, nrows, ncols, R, S)
            nb = EK.shape[0]
            bad = 0
            for b in range(nb):
                jb = min((b + 1) * R, nrows)
                if jb not in path:
                    continue                      # the path ended below this band
                kin = path[jb]
                hi = np.searchsorted(starts, kin)
                lo = hi if starts[hi] == kin else hi - 1
                if not (EK[b, lo] == EK[b, hi] and EJ[b, lo] == EJ[b, hi] and EK[b, lo] > 0):
                    bad += 1
            line += "  R%d/S%d %d/%d(%d)" % (R, S, bad, nb, steps)
        print(line, flush=True)
        return 0

    return fill


name = sys.argv[1] if len(sys.argv) > 1 else "Set3"
nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 5
_, seqs = H.read_fasta(os.path.join(H.GOLDEN, "data", name + ".txt"))
seqs = seqs[:nseq]
rots = H.ref_rotations(seqs, timeout=120)[1]
print(name, [len(s) for s in seqs], rots)
fillfn = make_filler()
r = csa_amd.debug_align_with_filler((seqs, rots, None, None), fillfn)
print("status", r["status"], "consensus", r["consensus"])
