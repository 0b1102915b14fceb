#!/bin/bash
# Round 4, first look: where do the real pair batches (config 3) spend their time?  Kernel traces of the pipelined runs, the same
# with one stream, and a sweep of passes per launch x streams.   tools/r04/probe_real.sh gpurun_out/r04a
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d  fill alone %.3f  tb alone %.3f  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d["kernel_ms"]["fill_launch_alone"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
$B --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" config4 | tee -a $OUT/summary.txt
for w in mammals primates; do
  $B --mode strong --workload $w --steps 12 --warmup 3 2>/dev/null | python3 -c "$fmt" $w | tee -a $OUT/summary.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${w}_stats -- $B --mode strong --workload $w --steps 12 --warmup 3 > "$OUT/log_${w}_stats.txt" 2>&1
  CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${w}_solo -- $B --mode strong --workload $w --steps 12 --warmup 3 > "$OUT/log_${w}_solo.txt" 2>&1
done
for sg in "2 1" "2 2" "2 3" "2 4" "3 2" "3 3" "4 2" "4 3" "4 4"; do set -- $sg
  for w in mammals primates; do
    CSADP_BITS_GROUP=$2 CSADP_BITS_STREAMS=$1 $B --mode strong --workload $w --steps 12 --warmup 3 2>/dev/null | python3 -c "$fmt" "$w streams $1 group $2" | tee -a $OUT/summary.txt
  done
done
ls $OUT
