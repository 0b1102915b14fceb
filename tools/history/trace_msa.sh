#!/bin/bash
# rocprofv3 kernel trace of the alignment-stage probe on one example set; prints the kernel timeline of the last call
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SET=${1:-Set3}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/msa_trace
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/msa_trace -o t -- python3 $ROOT/tools/msa_probe.py $SET > $ROOT/gpurun_out/msa_trace.log 2>&1
tail -3 $ROOT/gpurun_out/msa_trace.log
python3 - <<PY
import csv, glob
fn = glob.glob('$ROOT/gpurun_out/msa_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(len(rows)/3):]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-28s start %9.1f us  dur %8.1f us  gap-before %8.1f us  grid %s" % (r['Kernel_Name'][:28].replace('void csadp::',''), (st-t0)/1e3, (en-st)/1e3, (st-prev_end)/1e3, r.get('Grid_Size_X', r.get('Grid_Size',''))))
    prev_end = en
PY
