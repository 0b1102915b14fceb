#!/usr/bin/env python3
"""Build gate for the hand-counted `s_waitcnt vmcnt(young)` at the head of nw_fill_cells' generated block (tools/gen_cells_block.py).

The statement asks for the NEXT block's letters at its step 18 (two global_load_dwordx4) and waits for them at the next block's head with
vmcnt(young), young = the vector memory instructions the wave issues in between (csadp_cells.hip: 4 direction stores, + 1 granule store in
the plain layout's last strip, + 1 granule request in a chunk's first strip).  Inline assembly is opaque to the compiler: should it ever
place one more vector memory instruction on that path (a spilled register, a job field re-read as a vector load, a probe's timer store),
the letters would still be in flight at the wait and the fill silently wrong.  This script walks the compiled ISA: for every generated
block (an asm statement of > 300 lines) it finds the loop the block sits in (the first branch behind the block that jumps to a label in
front of it) and counts the vector memory instructions on the cycle from behind the letter loads, through the loop's back edge, to the
block's head.  The counts must equal EXPECTED -- taken from the build the GPU parity suite ran green on -- else the build fails and
`young` has to be derived again.

usage: check_cells_vmcnt.py <csadp_cells.isa>
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)(load|store|atomic)")
BRANCH = re.compile(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_ZN5csadp13nw_fill_cellsILb([01])ELb([01])EE\w+):")

# (wide, fetch) -> sorted list of the cycles' counts, one per generated block of the kernel.
# Layout with helper waves (fetch = 1): roles FIRST and RING only: 4 direction stores.  Plain layout: FIRST, RING with the publishing store
# under a branch (counted: 5), CHUNK unrolled by two (each copy: 4 stores + the request + the publishing store).
EXPECTED = None   # filled in below (kept at the end of the file so that the table is easy to find and to update)


def count_vmem(lines, path):
    n = 0
    nested = False
    for k in path:
        if "ASMSTART" in lines[k]:
            nested = True
        if VMEM.match(lines[k]):
            # granule_reload (the rare second look) carries its own `s_waitcnt vmcnt(0)`: everything is drained behind it
            if not (nested and k + 1 < len(lines) and "vmcnt(0)" in lines[k + 1]):
                n += 1
        if "ASMEND" in lines[k]:
            nested = False
    return n


def cycles(lines, lo, hi):
    labels = {}
    for i in range(lo, hi):
        m = LABEL.match(lines[i])
        if m:
            labels[m.group(1)] = i
    blocks = []                      # (first line, last line) of every generated block of the kernel
    i = lo
    while i < hi:
        if "ASMSTART" in lines[i]:
            j = i
            while "ASMEND" not in lines[j]:
                j += 1
            if j - i > 300:
                blocks.append((i, j))
            i = j
        i += 1
    out = []
    for bi, (i, j) in enumerate(blocks):
        loads = [k for k in range(i, j) if "global_load_dwordx4" in lines[k]]
        if len(loads) < 2:
            raise SystemExit("check_cells_vmcnt: a generated block without its two letter loads at line %d" % (i + 1))
        behind = loads[-1] + 1
        nxt = blocks[bi + 1][0] if bi + 1 < len(blocks) else hi
        back = None
        for k in range(j, nxt):
            m = BRANCH.match(lines[k])
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                back = (k, labels[m.group(1)])
                break
        if back is None:
            if bi + 1 == len(blocks):
                raise SystemExit("check_cells_vmcnt: no back edge behind the block at line %d" % (i + 1))
            path = list(range(behind, nxt))           # an unrolled loop: the next block follows in line
        else:
            head = next(b[0] for b in blocks if b[0] >= back[1])     # the first block behind the loop's label (this one, or its unrolled partner)
            path = list(range(behind, back[0])) + list(range(back[1], head))
        out.append(count_vmem(lines, path))
    return sorted(out)


def main():
    lines = open(sys.argv[1]).read().splitlines()
    funcs = []
    for i, ln in enumerate(lines):
        m = FUNC.match(ln)
        if m:
            funcs.append((i, int(m.group(2)), int(m.group(3))))
    if len(funcs) != 4:
        raise SystemExit("check_cells_vmcnt: expected the four nw_fill_cells kernels, found %d" % len(funcs))
    got = {}
    for n, (start, wide, fetch) in enumerate(funcs):
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        got[(wide, fetch)] = cycles(lines, start, end)
    if "--print" in sys.argv:
        print(got)
        return
    bad = {k: (got[k], EXPECTED[k]) for k in EXPECTED if got.get(k) != EXPECTED[k]}
    if bad:
        for k, (g, e) in bad.items():
            print("nw_fill_cells<wide=%d, fetch=%d>: vector memory instructions between a block's letter loads and the next block's head: %s, expected %s"
                  % (k[0], k[1], g, e))
        print("-> the hand-counted vmcnt at the head of the generated block (tools/gen_cells_block.py, `young` in csadp_cells.hip) no longer matches the code around it")
        raise SystemExit(1)
    print("%d kernel(s): the letter loads' wait count matches the code between two blocks" % len(got))


EXPECTED = {
    (1, 1): [4, 4], (0, 1): [4, 4],                 # helper-wave layout: roles FIRST, RING: the four direction stores
    (1, 0): [4, 4, 6, 6], (0, 0): [4, 4, 6, 6],     # plain layout: FIRST, RING (their publishing store sits out of line), CHUNK unrolled by two:
                                                    # 4 stores + the request + the publishing store under a branch no first strip ever takes (young = 5)
}

if __name__ == "__main__":
    main()
