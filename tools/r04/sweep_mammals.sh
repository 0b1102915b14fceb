#!/bin/bash
# Mammals (66 pairs, three strips of three words): streams x passes per launch x LDS reservation of the fill workgroups, 20 and 48 steps
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd $ROOT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d.get("verified")))'
for w in mammals; do
for pad in 0 24 36; do
for sg in "4 3" "4 2" "3 3" "4 4" "2 4" "4 5"; do set -- $sg
    for st in "20 5" "48 8"; do set -- $sg $st
    CSADP_BITS_LDS_PAD=$pad CSADP_BITS_GROUP=$2 CSADP_BITS_STREAMS=$1 $B --mode strong --workload $w --steps $3 --warmup $4 2>/dev/null | python3 -c "$fmt" "$w $3/$4 pad $pad streams $1 group $2" | tee -a $OUT/summary.txt
    done
done
done
done
