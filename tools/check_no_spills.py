#!/usr/bin/env python3
"""Build gate for kernels that hide an outstanding load from the compiler (nw_fill_cells requests hand-off granules with inline
assembly two blocks ahead, csadp_cells.hip: granule_request): the destination register pair must stay where it is until the wait, so
the kernel may neither spill vector registers nor use scratch memory -- a spilled or re-materialised copy would be taken before
the data lands.  Usage: check_no_spills.py file.isa KERNEL_SUBSTRING...   (reads the .amdgpu_metadata of a -S compile)"""
import re
import sys

text = open(sys.argv[1]).read()
bad = 0
seen = 0
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", text, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s*(\S+)", blk)
    if not name or not any(k in name.group(1) for k in sys.argv[2:]):
        continue
    seen += 1
    priv = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1))
    vsp = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", blk).group(1))
    if priv or vsp:
        print("SPILL: %s uses %d bytes of scratch, %d spilled vector registers" % (name.group(1), priv, vsp))
        bad += 1
print("%d kernel(s) checked, %d with vector spills or scratch" % (seen, bad))
sys.exit(1 if bad or not seen else 0)
