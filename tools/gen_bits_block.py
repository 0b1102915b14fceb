#!/usr/bin/env python3
"""The steady-state block of 32 steps of nw_fill_bits (csadp_bits.hip) as ONE inline-assembly statement.

    python tools/gen_bits_block.py csa_amd/csrc/csadp_bits_block.inc          what ships: W = 1, for the one-wave-per-SIMD launches
    python tools/gen_bits_block.py build/csadp_bits_block.inc --probe         every variant, W = 1 and 2, for tools/subco_probe.hip

Measured (round 3).  Many waves per SIMD: in the probe the block runs 4-5 % faster than the C++ form (181 / 167 / 160 against
190 / 175 / 166 cycles per step of two words at 2 / 4 / 8 waves per SIMD), built into nw_fill_bits it was no faster
(profiles/r03_ab_asm_block.txt), and the same block at 0 instead of 4 modulo 8 bytes runs 14 % slower: the many-wave kernels keep the
C++ form.  ONE wave per SIMD (a single matrix): the preference flips -- at 0 mod 8 the block takes 150 cycles per step against the
C++ form's 161 -- and in the kernel a 16 kbp pair fills in 1.43 instead of 1.59 ms, a 200 kbp pair in 17.3 instead of 18.6 ms
(profiles/r03_single_probe.txt): shipped for those launches.

Why.  The C++ form of the step (bits_block) keeps its DPP / carry instructions in small `asm volatile` statements (the carry
must stay in VCC between them).  The compiler treats every such statement as an instruction of unknown kind: it puts an
`s_nop 0` between the statement and the first instruction that reads one of its outputs -- three per step, always, because
everything downstream of a chain reads it -- and it may not move anything across the volatile statements, so the two words
of a lane are not interleaved the way their data flow allows.  Here the whole block is one statement: no s_nop, the order
below is the order executed, the LDS reads of the next step's first-lane inputs are issued right after the last read of the
registers they overwrite, and every wait is a counted s_waitcnt of the statement's own LDS reads.

Operands (see asm_block in tools/subco_probe.hip): the lane state (+v), the letter-chain constants and the LDS byte address
of this lane's input rows (v).  Temporaries are fixed VGPRs v[T0 .. T0+NT), declared as clobbers.

Register use per step t (W = 2):   load set L[t % 2] = {P0, P1, Z2, Z1 | Z0}: x0 / x1 of the step live in P0 / P1 of that set
until the next step has chained from them; the other set is re-loaded for step t + 1 right after this step's two v_xor_b32_dpp.

"""
import sys

LA, LB, LC = 0xF0, 0xCC, 0xAA
T0 = 40                 # first temporary VGPR
DPP = "wave_shr:1 row_mask:0xf bank_mask:0xf"
INJ_STRIDE = 32         # bytes per step in the inject / constant rows (kInjWords * 4)


def tab(expr):
    return expr & 0xff


class Regs:
    """names -> 'v<n>' for the fixed temporaries, '%[name]' for operands"""

    def __init__(self, W):
        self.W = W
        n = T0
        self.fixed = {}
        # two load sets, each 4 consecutive registers (ds_read_b128; even-aligned tuples) + 1
        for s in range(2):
            n = (n + 3) // 4 * 4
            self.fixed["L%d" % s] = n
            n += 4
            self.fixed["Z0_%d" % s] = n
            n += 1
        for nm in ["JUNK", "TX"]:
            self.fixed[nm] = n
            n += 1
        for h in range(W):
            for nm in ["NE", "GT", "S", "G2", "T1", "G1T", "A1", "G1", "O0", "G0", "C1"]:
                self.fixed["%s%d" % (nm, h)] = n
                n += 1
            # registers whose first value is dead by the time the second one is written
            self.fixed["V%d" % h] = self.fixed["T1%d" % h]      # t1 is last read by g1
            self.fixed["WW%d" % h] = self.fixed["T1%d" % h]     # w is written in place of v
            self.fixed["B0%d" % h] = self.fixed["T1%d" % h]     # ... and is last read by O0
            self.fixed["C0%d" % h] = self.fixed["GT%d" % h]     # g2 is last read by the complemented outgoing plane
            self.fixed["AA%d" % h] = self.fixed["S%d" % h]      # the chain's sum is last read by G1
        self.end = n

    def t(self, name):
        return "v%d" % self.fixed[name]

    def quad(self, s):
        b = self.fixed["L%d" % s]
        return "v[%d:%d]" % (b, b + 3)

    def lp(self, s, i):      # register i (0: P0, 1: P1, 2: Z2, 3: Z1) of load set s
        return "v%d" % (self.fixed["L%d" % s] + i)


def bitop(dst, a, b, c, expr):
    return "v_bitop3_b32 %s, %s, %s, %s bitop3:0x%02x" % (dst, a, b, c, tab(expr))


def block(W, all8=False, phase=0, paired=True, early=False):
    """phase: 0 or 4 = where the block starts modulo 8 bytes (after a .p2align 3).  paired: the 4-byte instructions come in
    adjacent pairs, so that every 8-byte instruction of the block starts at the same offset modulo 8 (the few 4-byte-capable
    instructions without a partner take their 8-byte encoding)."""
    R = Regs(W)
    addc = "v_addc_co_u32_e64 %s, vcc, %s, %s, vcc" if all8 else "v_addc_co_u32 %s, vcc, %s, %s, vcc"
    addc8 = "v_addc_co_u32_e64 %s, vcc, %s, %s, vcc"
    vor = "v_or_b32_e64 %s, %s, %s" if all8 else "v_or_b32 %s, %s, %s"
    vxor = "v_xor_b32_e64 %s, %s, %s" if all8 else "v_xor_b32 %s, %s, %s"
    a = []
    op = lambda name: "%%[%s]" % name
    nh0 = [op("nh0_%d" % h) for h in range(W)]
    h1 = [op("h1_%d" % h) for h in range(W)]
    h2 = [op("h2_%d" % h) for h in range(W)]
    a.append(".p2align 3")
    if phase == 4:
        a.append("s_nop 0")
    # nothing of the compiler's is outstanding on lgkmcnt inside the statement: its own waits are counted
    a.append("s_waitcnt lgkmcnt(0)")
    if paired:
        a.append("s_nop 0")                  # partner of the wait above
    a.append("ds_read_b128 %s, %s" % (R.quad(0), op("ip")))
    a.append("ds_read_b32 %s, %s offset:16" % (R.t("Z0_0"), op("ip")))
    for t in range(32):
        s, o = t % 2, (t + 1) % 2
        xprev0 = op("x0") if t == 0 else R.lp(o, 0)
        xprev1 = op("x1") if t == 0 else R.lp(o, 1)
        P0, P1, Z2, Z1, Z0 = R.lp(s, 0), R.lp(s, 1), R.lp(s, 2), R.lp(s, 3), R.t("Z0_%d" % s)
        if not (paired and t > 0):           # paired: the wait sits at the end of the step before, next to the last accumulator
            a.append("s_waitcnt lgkmcnt(0)")
            if paired:
                a.append("s_nop 0")
        a.append("v_xor_b32_dpp %s, %s, %s %s" % (P0, xprev0, op("d0"), DPP))
        a.append("v_xor_b32_dpp %s, %s, %s %s" % (P1, xprev1, op("d1"), DPP))
        if t < 31:           # the other set is free now: the previous step's x have been chained from, its carries used
            a.append("ds_read_b128 %s, %s offset:%d" % (R.quad(o), op("ip"), (t + 1) * INJ_STRIDE))
            a.append("ds_read_b32 %s, %s offset:%d" % (R.t("Z0_%d" % o), op("ip"), (t + 1) * INJ_STRIDE + 16))
        NE = [R.t("NE%d" % h) for h in range(W)]
        GT = [R.t("GT%d" % h) for h in range(W)]
        S = [R.t("S%d" % h) for h in range(W)]
        G2 = [R.t("G2%d" % h) for h in range(W)]
        T1 = [R.t("T1%d" % h) for h in range(W)]
        G1T = [R.t("G1T%d" % h) for h in range(W)]
        A1 = [R.t("A1%d" % h) for h in range(W)]
        G1 = [R.t("G1%d" % h) for h in range(W)]
        V = [R.t("V%d" % h) for h in range(W)]
        WW = [R.t("WW%d" % h) for h in range(W)]
        O0 = [R.t("O0%d" % h) for h in range(W)]
        G0 = [R.t("G0%d" % h) for h in range(W)]
        C1 = [R.t("C1%d" % h) for h in range(W)]
        C0 = [R.t("C0%d" % h) for h in range(W)]
        AA = [R.t("AA%d" % h) for h in range(W)]
        B0 = [R.t("B0%d" % h) for h in range(W)]
        sub = lambda no, z: "v_sub_co_u32_dpp %s, vcc, %s, %s %s bound_ctrl:0" % (R.t("JUNK"), op(no), z, DPP)
        if early:                            # early: each chain's borrow is in VCC long before the additions that take it
            a.append(sub("no2", Z2))
        # mismatch masks
        a.append(("v_or_b32_e64 %s, %s, %s" if paired and W == 1 else vor) % (NE[0], P0, P1))
        for h in range(1, W):
            a.append(vxor % (R.t("TX"), P0, op("e0_%d" % h)))
            a.append(bitop(NE[h], R.t("TX"), P1, op("e1_%d" % h), LA | (LB ^ LC)))
        for h in range(W):
            a.append(bitop(GT[h], NE[h], nh0[h], nh0[h], ~LA & LB))
        # chain ">= 2"
        if not early:
            a.append(sub("no2", Z2))
        for h in range(W):
            a.append((addc8 if paired and W % 2 else addc) % (S[h], nh0[h], GT[h]))
        if not early:
            a.append((addc8 if paired else addc) % (op("a2"), op("a2"), op("a2")))
        for h in range(W):
            a.append(bitop(G2[h], S[h], nh0[h], GT[h], LA ^ LB ^ LC))
        if early:
            a.append((addc8 if paired else addc) % (op("a2"), op("a2"), op("a2")))
            a.append(sub("no1", Z1))
        for h in range(W):
            a.append(bitop(T1[h], NE[h], nh0[h], G2[h], ~LA | (~LB & LC)))
        a.append(bitop(op("no2"), GT[W - 1], nh0[W - 1], G2[W - 1], ~(LA | (LB & LC))))
        for h in range(W):
            a.append(bitop(G1T[h], T1[h], h1[h], h1[h], LA & ~LB))
        for h in range(W):
            a.append(bitop(A1[h], G1T[h], NE[h], nh0[h], LA | (LB & LC)))
        # chain ">= 1"
        if not early:
            a.append(sub("no1", Z1))
        for h in range(W):
            a.append((addc8 if paired and W % 2 else addc) % (S[h], A1[h], G1T[h]))
        if not early:
            a.append((addc8 if paired else addc) % (op("a1"), op("a1"), op("a1")))
        for h in range(W):
            a.append(bitop(G1[h], S[h], A1[h], G1T[h], LA ^ LB ^ LC))
        if early:
            a.append((addc8 if paired else addc) % (op("a1"), op("a1"), op("a1")))
            a.append(sub("no0", Z0))
        for h in range(W):
            a.append(bitop(V[h], h1[h], G2[h], G1[h], (LA & LB) | (~LA & LC)))
        a.append(bitop(op("no1"), G1T[W - 1], A1[W - 1], G1[W - 1], ~(LA | (LB & LC))))
        for h in range(W):
            a.append(bitop(WW[h], NE[h], V[h], h2[h], ~LC & (~LA | LB)))
        for h in range(W):
            a.append(bitop(O0[h], WW[h], NE[h], nh0[h], LA | (LB & LC)))
        # plane ">= 0": G0 = 2 O0 + carry
        if not early:
            a.append(sub("no0", Z0))
        for h in range(W):
            a.append((addc8 if paired and W % 2 else addc) % (G0[h], O0[h], O0[h]))
        if not paired:
            a.append(addc % (op("a0"), op("a0"), op("a0")))
        for h in range(W):
            a.append(bitop(C1[h], NE[h], G2[h], h2[h], ~LA | LB | LC))
        a.append(bitop(op("no0"), O0[W - 1], O0[W - 1], O0[W - 1], ~LA))
        for h in range(W):
            a.append(bitop(C0[h], NE[h], G1[h], h1[h], ~LA | LB | LC))
        for h in range(W):
            a.append(bitop(h2[h], C1[h], G0[h], G0[h], LA & ~LB))
        for h in range(W):
            a.append(bitop(AA[h], C1[h], G1[h], G1[h], LA & ~LB))
        for h in range(W):
            a.append(bitop(B0[h], C0[h], G1[h], G0[h], LC & (~LA | LB)))
        for h in range(W):
            a.append(bitop(h1[h], G0[h], AA[h], C0[h], (LA & LB) | (~LA & LC)))
        for h in range(W):
            a.append(bitop(nh0[h], B0[h], C1[h], G2[h], LA & (~LB | LC)))
        if paired:                            # nothing since chain 0 has touched VCC: its last carry is still there
            a.append(addc % (op("a0"), op("a0"), op("a0")))
            a.append("s_waitcnt lgkmcnt(0)" if t < 31 else "s_nop 0")
    # x of the last step back into the operands the next block chains from
    a.append("v_mov_b32 %s, %s" % (op("x0"), R.lp(31 % 2, 0)))
    a.append("v_mov_b32 %s, %s" % (op("x1"), R.lp(31 % 2, 1)))
    return a, R


def cstring(lines):
    return " \\\n".join('\t"%s\\n\\t"' % l for l in lines)


def main():
    probe = "--probe" in sys.argv
    path = [x for x in sys.argv[1:] if not x.startswith("--")][0]
    out = ["/* GENERATED by tools/gen_bits_block.py -- do not edit.  See that file for the register map. */"]
    if not probe:
        # what ships: one word per lane, for the launches that run ONE wave per SIMD (a single matrix, the first fills of a
        # whole-genome profile alignment): 4-byte instructions paired, the block 8-byte aligned with its 8-byte instructions
        # at 0 mod 8 -- the placement a lone wave runs fastest (the many-wave kernels prefer 4 mod 8, and gain nothing from
        # the block: DESIGN.md section 3)
        lines, R = block(1, phase=0)
        out.append("#define BITS_BLOCK_ASM_W1_LONE \\\n%s" % cstring(lines))
        out.append("#define BITS_BLOCK_CLOBBERS_W1 " + ", ".join('"v%d"' % r for r in range(T0, R.end)) + ', "vcc", "memory"')
        steps = [l for l in lines if l.startswith("v_")]
        out.append("/* %d vector instructions per block = %.2f per step, temporaries v%d .. v%d */" % (len(steps), (len(steps) - 2) / 32.0, T0, R.end - 1))
        open(path, "w").write("\n".join(out) + "\n")
        return
    for W in (1, 2):
        # the probe's comparisons: 4-byte instructions paired and every 8-byte instruction at 4 mod 8, unpaired, early borrows, at 0 mod 8
        for name, kw in (("", dict(phase=4)), ("_PLAIN", dict(paired=False)), ("_EARLY", dict(phase=4, early=True)), ("_AT0", dict(phase=0))):
            lines, R = block(W, **kw)
            out.append("#define BITS_BLOCK_ASM_W%d%s \\\n%s" % (W, name, cstring(lines)))
        out.append("#define BITS_BLOCK_CLOBBERS_W%d " % W + ", ".join('"v%d"' % r for r in range(T0, R.end)) + ', "vcc", "memory"')
        steps = [l for l in lines if l.startswith("v_")]
        out.append("/* W = %d: %d vector instructions per block = %.2f per step, temporaries v%d .. v%d */" % (W, len(steps), (len(steps) - 2) / 32.0, T0, R.end - 1))
    open(path, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
