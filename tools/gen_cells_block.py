#!/usr/bin/env python3
"""Generates csa_amd/csrc/csadp_cells_block.inc: the hand-scheduled 32-step block of nw_fill_cells as ONE
inline-assembly statement per variant, on fixed VGPRs (so that lane 63's hand-off values sit in register
tuples for 16-byte LDS stores and nothing depends on the compiler's register allocation).

A lane owns TWO adjacent columns, A (even) and B (odd), and computes both cells of a row in one step: A takes its
left neighbour (the left lane's B) through DPP, B takes A's fresh value from a register.

Instruction placement: a step is 84 bytes (five 4-byte instructions), so every second step has its 8-byte instructions at 4 mod 8.  The
bit-parallel block gains 15 % from keeping them at 0 mod 8 (tools/gen_bits_block.py); this one gains nothing (round 4: one 4-byte
instruction per step in its VOP3 form behind a .p2align 3 -- a strip alone 90 against 91 cycles per step, a 16 384^2 fill 1.156 against
1.153 ms) and stays as it is.

Register map (clobbered by the statement; BASE = 56: the compiler's own values of the kernel -- 52 registers -- fit below it, so the kernel
takes 150 registers and a SIMD holds THREE of its waves.  Rounds 2-4 had the map at v128 and 32 registers for OXB: 254 registers, two waves):
  v[BASE:BASE+35]      XW   the 9 x ds_read_b128 window of hand-off values; word 3 + t = XP(t), the lane-0 preset of
                            step t (hand-off value + leftcA), which v_add_u32_dpp turns into lfA(t) for all other lanes
  v[BASE+36:BASE+43]   OXB  X of the lane's B cell after step t, in register t % 8 (what lane 63 hands to the next strip: a 16-byte LDS store
                            of four of them every four steps, from the half of the eight that the next four steps do not write)
  v[BASE+44:BASE+75]   Y    letter offset of step t (lane 0 preset from the row sequence, v_mov_b32_dpp for the others)
  v[BASE+76:BASE+83]   H, G, ACCA, ACCB, DGA, DGB, OXA, LFB
  v(BASE+84)           LM   ramp variants (a strip's first two blocks): -4 in the lanes whose first row has arrived, 0 in the others; moves one
                   lane to the right per step like the letters.  A lane that is not live yet runs the same instructions on its border
                   values and keeps them: X is taken through v_bfi_b32 (LM ? h & -4 : old X) where the later blocks have v_and_b32;
                   everything else such a lane computes is either unused or already what its first row needs (lf = the left lane's border
                   value + leftc = its own D; the next step's dg from that and the letter that arrives with the row)
  v[BASE+86:BASE+93]   LT   the block's 32 letter offsets (below)
"""
import os
import sys

BASE = int(os.environ.get("CELLS_BLOCK_BASE", "56"))      # (128: the occupancy of rounds 2-4, two waves per SIMD, for A/B runs)
XW, OXB0, Y = BASE, BASE + 36, BASE + 44
H, G, ACCA, ACCB, DGA, DGB, OXA, LFB, LM = (BASE + 76 + i for i in range(9))
LT = BASE + 86  # eight registers (tuples start at even registers): the block's 32 letter offsets, the same in every lane (operands of the statement, pinned by register variables)
PX = XW + 3
DPP = "wave_shr:1 row_mask:0xf bank_mask:0xf"


def OXB(t):
    return OXB0 + t % 8


def gain(wide, ysrc, col):
    if wide:
        return ["v_bfe_u32 v%d, %%[tab%s], v%d, 6" % (G, col, ysrc), "v_lshl_add_u32 v%d, v%d, 3, %%[c2%s]" % (G, G, col)]
    return ["v_bfe_u32 v%d, %%[tab%s], v%d, 8" % (G, col, ysrc)]


def ring_poll(need, lo, hi, tag):
    """ROLE_RING, the slow way to a half of the window: bounded poll of the producer's half-block counter (in LDS: %[paddr], the same address in
    every lane), then the window's words again.  Only taken when the look that was asked for ahead of time (counter first, words behind it: the
    LDS runs a wave's reads in order) found the counter short.  %[tmo] = 1: the poll ran out (the caller raises the abort word; the values
    read are then garbage)."""
    a = ["s_waitcnt lgkmcnt(0)",
         "v_readfirstlane_b32 %[sval], %[vtmp]",
         "s_cmp_ge_i32 %[sval], " + need,
         "s_cbranch_scc1 %d2f" % tag,
         "s_mov_b32 %[scnt], 0x400000",
         "%d1:" % tag,
         "ds_read_b32 %[vtmp], %[paddr]",
         "s_waitcnt lgkmcnt(0)",
         "v_readfirstlane_b32 %[sval], %[vtmp]",
         "s_cmp_ge_i32 %[sval], " + need,
         "s_cbranch_scc1 %d3f" % tag,
         "s_sub_u32 %[scnt], %[scnt], 1",
         "s_cmp_lg_u32 %[scnt], 0",
         "s_cbranch_scc1 %d1b" % tag,
         "s_mov_b32 %[tmo], 1",
         "%d3:" % tag]
    a += ring_words(lo, hi)
    a += ["s_waitcnt lgkmcnt(0)", "%d2:" % tag]
    return a


def ring_words(lo, hi):
    """the window's 16-byte pieces lo..hi-1: the first comes from %[raddr], the rest from %[raddrb] (+ 16 q): where the 36 words straddle the
    ring's end the second address is the ring's start - 16 (round 3 mirrored the ring's first two blocks behind its end instead)"""
    a = []
    for q in range(lo, hi):
        if q == 0:
            a.append("ds_read_b128 v[%d:%d], %%[raddr]" % (XW, XW + 3))
        else:
            a.append("ds_read_b128 v[%d:%d], %%[raddrb] offset:%d" % (XW + 4 * q, XW + 4 * q + 3, 16 * q))
    return a


def block(wide, role, ramp=False):
    first = role == "FIRST"
    tag = 3 if ramp else 1                 # numeric labels of the polls: the two bodies of a statement keep theirs apart
    a = []
    if ramp:
        a.append("v_mov_b32 v%d, %%[lm0]" % LM)
    if first:
        # the job's first strip: hand-off value of row r = border column X[r][0] = leftmul * r (:967); %[x0] = that of
        # the block's first row + leftcA, every further one adds leftmul
        a.append("v_mov_b32 v%d, %%[x0]" % PX)
        for t in range(1, 32):
            a.append("v_mad_i32_i24 v%d, %%[lm], %d, %%[x0]" % (PX + t, t))      # (a chain of 31 dependent adds before: +5 cycles per step)
    elif role == "RING":
        # the window in two halves: words 0..19 (the presets of steps 0..16) now, words 20..35 in front of step 17, each asked for together
        # with the producer's half-block counter (counter first) and looked at only after work that does not need it
        a.append("ds_read_b32 %[vtmp], %[paddr]")
        a += ring_words(0, 5)
    else:
        for q in range(9):
            a.append("ds_read_b128 v[%d:%d], %%[raddr] offset:%d" % (XW + 4 * q, XW + 4 * q + 3, 16 * q))
    # lane-0 letter offsets of the block's 32 rows: the bytes of eight scalar registers
    for t in range(32):
        a.append("v_bfe_u32 v%d, v%d, %d, 8" % (Y + t, LT + t // 4, 8 * (t % 4)))
    a.append("v_mov_b32 v%d, 0" % ACCA)
    a.append("v_mov_b32 v%d, 0" % ACCB)
    a.append("v_mov_b32 v%d, %%[outvA]" % OXA)
    # letter part of step 0 (every step does it for its successor)
    a.append("v_mov_b32_dpp v%d, %%[sh] %s" % (Y, DPP))
    a += gain(wide, Y, "A")
    a.append("v_add_u32 v%d, %%[dgA], v%d" % (DGA, G))
    a += gain(wide, Y, "B")
    a.append("v_add_u32 v%d, %%[dgB], v%d" % (DGB, G))
    # the window has had the time of the letter work to arrive
    if role == "RING":
        a += ring_poll("%[need1]", 0, 5, tag)
        for t in range(17):
            a.append("v_add_u32 v%d, v%d, %%[leftcA]" % (PX + t, PX + t))
    elif not first:
        a.append("s_waitcnt lgkmcnt(0)")
        for t in range(32):
            a.append("v_add_u32 v%d, v%d, %%[leftcA]" % (PX + t, PX + t))
    for t in range(32):
        prevB = "%[outvB]" if t == 0 else "v%d" % OXB(t - 1)
        if t == 12 and role == "RING":
            # the second half is asked for five steps before it is needed (the counter in front of it)
            a.append("ds_read_b32 %[vtmp], %[paddr]")
            a += ring_words(5, 9)
        if t == 18:
            # the next block's letters: asked for here (what the load overwrites was last read at the head of the block), waited for at the
            # head of the next statement -- every lane loads the same 32 bytes.  (Scalar loads share their counter with the LDS and return
            # out of order: the compiler waited lgkmcnt(0) three times per block for them; as loop-carried scalar operands of the
            # statement they do not compile -- the loop has divergent exits)
            a.append("global_load_dwordx4 v[%d:%d], %%[lvoff], %%[lbase]" % (LT, LT + 3))
            a.append("global_load_dwordx4 v[%d:%d], %%[lvoff], %%[lbase] offset:16" % (LT + 4, LT + 7))
        if t == 17 and role == "RING":
            a += ring_poll("%[need2]", 5, 9, tag + 1)
            for u in range(17, 32):
                a.append("v_add_u32 v%d, v%d, %%[leftcA]" % (PX + u, PX + u))
        if t < 31:
            a.append("v_mov_b32_dpp v%d, v%d %s" % (Y + t + 1, Y + t, DPP))
        else:
            a.append("s_nop 0")     # no letter move here: keep the X written two instructions ago two wait states away from its DPP read
        # column A: left neighbour = the left lane's B of this row
        a.append("v_add_u32_dpp v%d, %s, %%[leftcA] %s" % (PX + t, prevB, DPP))
        a.append("v_max3_i32 v%d, v%d, v%d, v%d" % (H, DGA, OXA, PX + t))
        if t < 31:
            a += gain(wide, Y + t + 1, "A")
            a.append("v_add_u32 v%d, v%d, v%d" % (DGA, PX + t, G))
        if ramp:
            a.append("v_bfi_b32 v%d, v%d, v%d, v%d" % (OXA, LM, H, OXA))
        else:
            a.append("v_and_b32 v%d, -4, v%d" % (OXA, H))
        a.append("v_alignbit_b32 v%d, v%d, v%d, 2" % (ACCA, H, ACCA))
        # column B: left neighbour = A, just computed
        a.append("v_add_u32 v%d, v%d, %%[leftcB]" % (LFB, OXA))
        a.append("v_max3_i32 v%d, v%d, %s, v%d" % (H, DGB, prevB, LFB))
        if t < 31:
            a += gain(wide, Y + t + 1, "B")
            a.append("v_add_u32 v%d, v%d, v%d" % (DGB, LFB, G))
        if ramp:
            a.append("v_bfi_b32 v%d, v%d, v%d, %s" % (OXB(t), LM, H, prevB))
            if t < 31:
                a.append("v_mov_b32_dpp v%d, v%d %s" % (LM, LM, DPP))
        else:
            a.append("v_and_b32 v%d, -4, v%d" % (OXB(t), H))
        a.append("v_alignbit_b32 v%d, v%d, v%d, 2" % (ACCB, H, ACCB))
        if t % 4 == 3:
            a.append("ds_write_b128 %%[waddr], v[%d:%d] offset:%d" % (OXB(t - 3), OXB(t), 16 * (t // 4)))
        if t == 15:
            # half of the block's hand-off values are in the ring: the half-block counter (lane 63's address is the counter,
            # every other lane's its scrap slot; the LDS runs a wave's stores in order)
            a.append("ds_write_b32 %[caddr], %[chalf]")
            a.append("v_mov_b32 %%[w0A], v%d" % ACCA)
            a.append("v_mov_b32 %%[w0B], v%d" % ACCB)
    # the block's hand-off values are in the ring: the counter 2 b + 2 (lane 63 of a strip that feeds one; scrap elsewhere), and for a ring
    # strip "block b of my producer's ring is in my registers" (lane 0; the producer may overwrite those slots) -- inside the statement:
    # they leave a hundred cycles earlier than behind the compiler's wait for the statement's LDS traffic, and without an EXEC dance each
    a.append("v_add_u32 %[vtmp], 1, %[chalf]")
    a.append("ds_write_b32 %[caddr], %[vtmp]")
    if role == "RING":
        a.append("ds_write_b32 %[taddr], %[tval]")
    a.append("v_mov_b32 %%[w1A], v%d" % ACCA)
    a.append("v_mov_b32 %%[w1B], v%d" % ACCB)
    a.append("v_mov_b32 %%[dgA], v%d" % (PX + 31))         # D of column A for the next block: lfA of the last step
    a.append("v_mov_b32 %%[dgB], v%d" % LFB)               # ... of column B: lfB
    a.append("v_mov_b32 %%[sh], v%d" % (Y + 31))
    a.append("v_mov_b32 %%[outvA], v%d" % OXA)
    a.append("v_mov_b32 %%[outvB], v%d" % OXB(31))
    return a


def cstring(lines):
    return " \\\n".join('\t"%s\\n\\t"' % l for l in lines)


out = ["/* GENERATED by tools/gen_cells_block.py -- do not edit.  See that file for the register map. */"]
def statement(wide, role):
    """one statement per (width, role): the first two blocks of a strip take the ramp body, all later ones the plain one -- ONE statement
    in the strip's loop, so that the lane state it carries from block to block stays in the same registers (two statements: a dozen copies
    per block at the loop's merge points)"""
    # the letters were asked for at step 18 of the previous block; since then this wave has issued %[young] = 4 (direction words) or 5 (+ a
    # hand-off granule store or request) vector memory instructions, which need not have finished
    head = ["s_cmp_eq_u32 %[young], 5", "s_cbranch_scc1 85f", "s_waitcnt vmcnt(4)", "s_branch 86f", "85:", "s_waitcnt vmcnt(5)", "86:"]
    return head + ["s_cmp_gt_u32 %[bidx], 1", "s_cbranch_scc0 70f"] + block(wide, role) + ["s_branch 80f", "70:"] + block(wide, role, True) + ["80:"]


for wide in (0, 1):
    for role in ("LDS", "FIRST", "RING"):
        out.append("#define CELLS_BLOCK_ASM_%s_%s \\\n%s" % ("WIDE" if wide else "BYTE", role, cstring(statement(wide, role))))
regs = list(range(XW, XW + 36)) + list(range(OXB0, OXB0 + 8)) + list(range(Y, Y + 32)) + [H, G, ACCA, ACCB, DGA, DGB, OXA, LFB, LM]
out.append("#define CELLS_BLOCK_CLOBBERS " + ", ".join('"v%d"' % r for r in regs) + ', "memory"')
for i in range(8):
    out.append('#define CELLS_LT%d "v%d"' % (i, LT + i))
open(sys.argv[1], "w").write("\n".join(out) + "\n")
