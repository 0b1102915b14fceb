#!/bin/bash
# rocprofv3 evidence for the profile path (mode N of the three example sets through csa_msa's library entry): kernel stats, then PMC passes
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04msa
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROBE="python3 $ROOT/tools/msa_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o msa_stats -- $PROBE > "$OUT/log_stats.txt" 2>&1
for C in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"; do
	TAG=$(echo $C | cut -d' ' -f1)
	rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o msa_pmc_$TAG -- $PROBE Set3 > "$OUT/log_$TAG.txt" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/**/msa_pmc_*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("csadp::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
summ = {}
for k, v in acc.items():
    if "nw_" not in k: continue
    d = dict(v)
    d["lds_conflict_cycles_per_lds_inst"] = round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_INSTS_LDS"], 1), 3)
    d["wait_any_over_wave_cycles"] = round(v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 3)
    d["active_inst_any_over_wave_cycles"] = round(v["SQ_ACTIVE_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 3)
    d["valu_insts_per_wave"] = round(v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), 1)
    summ[k] = d
    print(k, json.dumps({a: d[a] for a in ("lds_conflict_cycles_per_lds_inst", "wait_any_over_wave_cycles", "active_inst_any_over_wave_cycles", "valu_insts_per_wave")}))
json.dump(summ, open(out + "/msa_pmc_summary.json", "w"), indent=1, sort_keys=True)
PY
find "$OUT" -name "msa_stats_kernel_stats.csv" | xargs -n1 head -12
