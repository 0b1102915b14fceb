#!/bin/bash
# the profile-path tests, the example sets' DP times and per-launch kernel durations after a change to the traceback kernels
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r04b}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_msa.py tests/test_gpu_dropin.py tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest.log" 2>&1
rc=$?
tail -3 "$OUT/pytest.log"
[ $rc -lt 2 ] || exit $rc
timeout -k 10 300 python tools/msa_probe.py > "$OUT/msa_probe.log" 2>&1 || { tail -20 "$OUT/msa_probe.log"; exit 8; }
cat "$OUT/msa_probe.log"
bash tools/tb_cells_trace.sh Set3 > "$OUT/tbtrace_Set3.txt" 2>&1
grep scout "$OUT/tbtrace_Set3.txt" | awk '{s+=$2; n++} END {print "scout launches", n, "mean us", s/n}'
grep resolve "$OUT/tbtrace_Set3.txt" | awk '{s+=$2; n++} END {print "resolve launches", n, "mean us", s/n}'
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d "$OUT" -o msa_pmc_lds -- python3 $ROOT/tools/msa_probe.py Set3 > "$OUT/log_msa_pmc.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/msa_pmc_lds_counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0][-24:]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if v.get("SQ_INSTS_LDS"):
            print(k, "LDS conflict cycles per LDS instruction: %.2f" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_INSTS_LDS"]))
PY
