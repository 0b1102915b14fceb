#!/bin/bash
# Steady state of the real pair batches: streams x passes per launch x LDS reservation of the fill workgroups, 48 timed steps.
#   tools/r04/sweep_real.sh gpurun_out/r04b
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/$1
mkdir -p "$OUT"
cd $ROOT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d  fill alone %.3f  tb alone %.3f  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d["kernel_ms"]["fill_launch_alone"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
for w in mammals primates; do
  $B --mode strong --workload $w --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "$w default 48/8" | tee -a $OUT/summary.txt
  $B --mode strong --workload $w --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "$w default 20/5" | tee -a $OUT/summary.txt
done
for pad in 0 16 28; do
for sg in "2 3" "2 4" "3 3" "4 2" "4 3" "4 4"; do set -- $sg
  for w in mammals primates; do
    CSADP_BITS_LDS_PAD=$pad CSADP_BITS_GROUP=$2 CSADP_BITS_STREAMS=$1 $B --mode strong --workload $w --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "$w pad $pad streams $1 group $2" | tee -a $OUT/summary.txt
  done
done
done
