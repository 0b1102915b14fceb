#!/bin/bash
# per-launch durations of the profile-step kernels on one example set (GPU box): tools/tb_cells_trace.sh Set3
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/tbtrace_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 $ROOT/tools/msa_probe.py $1 > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "cells" in n or "nw_tb" in n:
        out.append("%s %.1f us grid %s" % (n.split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
open(sys.argv[1] + "/launches.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[-80:]))
PY
tail -3 "$OUT/log.txt"
