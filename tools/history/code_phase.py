#!/usr/bin/env python3
"""Where do the 8-byte instructions of a kernel's unrolled step block sit?  On gfx950 the same block runs 6 % faster or slower
depending on whether most of its 8-byte instructions start at 4 mod 8 or at 0 mod 8 (measured on nw_fill_bits: one s_nop in front
of the block loop: 0.87 -> 0.92 ms per pass; two: 0.87 again).  Prints, per kernel of a .hip file, every
branch-free block of 600 instructions or more and the share of its 8-byte instructions that start at 4 mod 8.

    python tools/code_phase.py [csadp_bits.hip]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    # --warn KERNEL MIN: a line on stderr for every block of a kernel whose name contains KERNEL with less than MIN percent
    warn = None
    if "--warn" in sys.argv:
        i = sys.argv.index("--warn")
        warn = (sys.argv[i + 1], float(sys.argv[i + 2]))
        del sys.argv[i:i + 3]
    src = sys.argv[1] if len(sys.argv) > 1 else "csadp_bits.hip"
    with tempfile.TemporaryDirectory() as tmp:
        obj, co = os.path.join(tmp, "k.o"), os.path.join(tmp, "k.co")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "--cuda-device-only", "-c", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "csa_amd", "csrc"),
                               os.path.join(ROOT, "csa_amd", "csrc", src), "-o", obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co]).decode().splitlines()
    heads = [(i, l) for i, l in enumerate(dis) if re.match(r"^[0-9a-f]+ <", l)]
    for n, (i, l) in enumerate(heads):
        j = heads[n + 1][0] if n + 1 < len(heads) else len(dis)
        ins = []
        for x in dis[i + 1:j]:
            m = re.match(r"\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", x)
            if m:
                ins.append((int(m.group(2), 16), m.group(1).split()[0]))
        blocks, cur = [], []
        for k in range(len(ins) - 1):
            a, op = ins[k]
            cur.append((a, op, ins[k + 1][0] - a))
            if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm":
                blocks.append(cur)
                cur = []
        if cur:
            blocks.append(cur)
        if not blocks:
            continue
        name = l.split("<")[1].rstrip(">:")
        for big in blocks:
            if len(big) < 600:              # the unrolled 32-step blocks have 800 .. 1400 instructions
                continue
            n8 = sum(1 for a, op, sz in big if sz == 8)
            odd = sum(1 for a, op, sz in big if sz == 8 and a % 8)
            print("%-70s block of %4d instructions at 0x%x: %4d of %4d 8-byte instructions at 4 mod 8 (%.0f %%)"
                  % (name[:70], len(big), big[0][0], odd, n8, 100.0 * odd / max(n8, 1)))
            if warn and warn[0] in name and 100.0 * odd < warn[1] * max(n8, 1):
                sys.stderr.write("WARNING: %s: a step block is in the SLOW code phase (%.0f %% of its 8-byte instructions at 4 mod 8): "
                                 "one s_nop in front of the block loop moves it\n" % (name[:60], 100.0 * odd / max(n8, 1)))


if __name__ == "__main__":
    main()
