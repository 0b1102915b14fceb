#!/bin/bash
# rocprofv3 evidence for the real pair batch (config 3, Mammals 66 pairs, three words per lane): kernel stats pipelined and one launch at a time, PMC passes
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r04mam
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs --mode strong --workload mammals"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o mam_stats -- $BENCH --steps 24 --warmup 4 > "$OUT/log_stats.txt" 2>&1
CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o mam_solo -- $BENCH --steps 12 --warmup 3 > "$OUT/log_solo.txt" 2>&1
for C in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"; do
	TAG=$(echo $C | cut -d' ' -f1)
	CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o mam_pmc_$TAG -- $BENCH --steps 6 --warmup 0 > "$OUT/log_$TAG.txt" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + "/**/mam_pmc_*_counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("csadp::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen and r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAIT_ANY"):
            seen.add((k, r["Dispatch_Id"]))
for k, v in acc.items():
    if "nw_" not in k: continue
    d = dict(v)
    d["lds_conflict_cycles_per_lds_inst"] = round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_INSTS_LDS"], 1), 3)
    d["wait_any_over_wave_cycles"] = round(v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 3)
    d["valu_insts_per_wave"] = round(v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), 1)
    print(k, json.dumps({a: d[a] for a in ("lds_conflict_cycles_per_lds_inst", "wait_any_over_wave_cycles", "valu_insts_per_wave")}))
json.dump({k: dict(v) for k, v in acc.items()}, open(out + "/mam_pmc_summary.json", "w"), indent=1, sort_keys=True)
PY
find "$OUT" -name "*kernel_stats.csv" | xargs -n1 head -6
