#!/bin/bash
# nw_fill_cells in chip-filling launches (256 families of 8 x 4 kbp, two round groups): the plain layout (two workgroups of four waves per compute unit)
# against the helper-wave layout forced on (one workgroup of six waves per unit): kernel durations and the SQ counters that say why
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r05_cells_layout
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CSADP_ROUND_GROUPS=2
P="python3 $ROOT/tools/r05/profile_batch_probe.py 256x8x4000"
for lay in plain:0 helper:100000; do
  name=${lay%%:*}; val=${lay##*:}
  CSADP_CELLS_FETCH=$val rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${name}_stats -- $P > $OUT/log_${name}_stats.txt 2>&1
  CSADP_CELLS_FETCH=$val rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o ${name}_pmc_a -- $P > $OUT/log_${name}_a.txt 2>&1
  CSADP_CELLS_FETCH=$val rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS -d $OUT -o ${name}_pmc_b -- $P > $OUT/log_${name}_b.txt 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for name in ("plain", "helper"):
    st = glob.glob(os.path.join(out, "**", name + "_stats_kernel_stats.csv"), recursive=True)[0]
    for r in csv.DictReader(open(st)):
        if "nw_fill_cells" in r["Name"]:
            print("%-6s %s: %s launches, average %.1f us (min %.1f, max %.1f)" % (name, r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    c = collections.Counter()
    for f in glob.glob(os.path.join(out, "**", name + "_pmc_*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "nw_fill_cells" in r["Kernel_Name"]:
                c[r["Counter_Name"]] += float(r["Counter_Value"])
    wc = c["SQ_WAVE_CYCLES"]
    print("%-6s per wave: %.0f VALU instructions, %.0f cycles x 4; of the wave cycles: issuing %.3f (VALU %.3f, LDS %.3f), waiting %.3f (for an instruction to be fetched or issued %.3f)" % (
        name, c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 4 * wc / c["SQ_WAVES"], c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_LDS"] / wc,
        c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc))
PY
