#!/usr/bin/env python3
"""ISA count of a kernel's steady-state block: compiles one .hip file for gfx950 to assembly (device only),
takes the largest basic block of the named kernel that holds no v_cndmask -- for nw_fill_bits /
nw_fill_cells that is the fully unrolled block of 32 steady-state steps (the ramp variant, which keeps
lanes above the matrix idle with v_cndmask, is one of the first two blocks of a strip only) -- and counts
its instructions by kind.

    python tools/count_valu.py [csadp_bits.hip nw_fill_bitsILi2ELi4ELb0] [--steps 32]     (W = 2, 4 waves, one workgroup per job)

Prints the VALU instructions per step and the mix bench.py prices (v_bitop3 / three-operand and DPP /
two-operand).  Needs hipcc only (no GPU)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF_RATE = ("v_addc_co", "v_sub_co", "v_add_co", "v_add3", "v_perm", "v_bfe", "v_alignbit", "v_max3", "v_min3", "v_max_", "v_min_", "v_lshl_add", "v_lshl_or",
             "v_and_or", "v_or3", "v_bfi", "v_cndmask", "v_cmp", "v_lshlrev", "v_mad", "v_mul", "v_readlane", "v_readfirstlane")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src = args[0] if args else "csadp_bits.hip"
    kernel = args[1] if len(args) > 1 else "nw_fill_bitsILi2ELi4ELb0"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 32
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "csa_amd", "csrc"),
                               os.path.join(ROOT, "csa_amd", "csrc", src), "-o", out])
        text = open(out).read()
    m = re.search(r"^(_ZN5csadp\d+%s\w*):.*?s_endpgm" % re.escape(kernel), text, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found in %s" % (kernel, src))
    blocks, cur = [], []
    for line in m.group(0).splitlines():
        ins = line.strip()
        if not ins or ins.startswith(";") or ins.startswith("."):
            if ins.startswith(".LBB") and cur:
                blocks.append(cur)
                cur = []
            continue
        cur.append(ins.split()[0])
        if ins.startswith("s_cbranch") or ins.startswith("s_branch"):
            blocks.append(cur)
            cur = []
    blocks.append(cur)
    steady = [b for b in blocks if not any(op.startswith("v_cndmask") for op in b)]
    body = max(steady or blocks, key=len)
    kinds = collections.Counter()
    for op in body:
        if op.startswith("v_"):
            if op.startswith("v_bitop3"):
                kinds["v_bitop3"] += 1
            elif "dpp" in op or op.startswith(HALF_RATE):
                kinds["dpp_or_three_operand"] += 1
            else:
                kinds["two_operand"] += 1
        elif op.startswith("ds_"):
            kinds["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            kinds["vmem"] += 1
        elif op.startswith("s_"):
            kinds["salu_or_wait"] += 1
    valu = kinds["v_bitop3"] + kinds["dpp_or_three_operand"] + kinds["two_operand"]
    print("%s %s: largest basic block %d instructions, %d VALU = %.2f per step of %d" % (src, m.group(1), len(body), valu, valu / steps, steps))
    for k, v in sorted(kinds.items()):
        print("  %-22s %5d  (%.2f per step)" % (k, v, v / steps))
    detail = collections.Counter(op for op in body if op.startswith("v_"))
    print("  " + ", ".join("%s x%d" % kv for kv in detail.most_common()))


if __name__ == "__main__":
    main()
