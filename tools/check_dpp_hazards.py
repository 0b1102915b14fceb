#!/usr/bin/env python3
"""The hand-scheduled steps of nw_fill_cells and the DPP instructions of nw_fill_bits / nw_traceback_replay are inline
assembly; the compiler's hazard recogniser does not look into it.  This checks the compiled ISA for the one rule they rely on: a VGPR written by a VALU
instruction is not read through DPP within the next two wait states (gfx9: 2).  Usage:
  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o cells.s csadp_cells.hip; check_dpp_hazards.py cells.s"""
import re
import sys

lines = [l.strip() for l in open(sys.argv[1])]
instrs = []          # (text, is_barrier) in order, per function; labels reset nothing (branches are conservative)
bad = 0
window = []          # last instructions: list of (written regs set, wait states it provides)


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


for ln in lines:
    if not ln or ln.startswith(";") or ln.startswith(".") or ln.endswith(":") or ln.startswith("//"):
        continue
    ln = ln.split(";")[0].strip()
    if not ln:
        continue
    parts = ln.replace(",", " ").split()
    op = parts[0]
    if op.endswith("_dpp") or "row_mask" in ln or "wave_shr" in ln or "row_shr" in ln or "quad_perm" in ln:
        # DPP source = the first source operand: after the destination, and after the carry-out of v_add_co / v_sub_co forms
        ops = [p for p in parts[1:] if p != "vcc"]
        src = regs_of(ops[1]) if len(ops) > 1 else set()
        dist = 0
        for wr, ws in reversed(window):
            if dist >= 2:
                break
            if wr & src:
                print("HAZARD: %s  (source written %d wait state(s) earlier)" % (ln, dist))
                bad += 1
                break
            dist += ws
    written = set()
    ws = 1
    if op == "s_nop":
        ws = int(parts[1]) + 1
    elif op.startswith("v_") and not op.startswith("v_cmp") and len(parts) > 1:
        written = regs_of(parts[1])
    window.append((written, ws))
    if len(window) > 8:
        window.pop(0)
print("%d DPP hazard(s)" % bad)
sys.exit(1 if bad else 0)
