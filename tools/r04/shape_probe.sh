#!/bin/bash
# Is a batch of three-strip workgroups slower per cell than one of four-strip workgroups?  Synthetic pairs of 12288 (3 strips of two words),
# 16384 (4 strips), 18000 letters (3 strips of three words), job counts chosen for equal waves per pass; 48 steps
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  W %d  passes/launch %d streams %d fill alone %.3f tb alone %.3f verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["words_per_lane"], d["config"]["passes_per_launch"], d["config"]["launches_in_flight"], d["kernel_ms"]["fill_launch_alone"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
for cfg in "16384 128" "12288 128" "12288 170" "12288 85" "18000 66" "18000 120" "18000 85" "18000 128" "18000 170" "16384 64" "16384 96"; do set -- $cfg
  $B --len $1 --pairs $2 --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "len $1 pairs $2 48/8" | tee -a $OUT/summary.txt
done
for cfg in "18000 66" "18000 128"; do set -- $cfg
  for sg in "2 2" "2 3" "4 2" "4 3" "3 2"; do set -- $cfg $sg
  CSADP_BITS_STREAMS=$3 CSADP_BITS_GROUP=$4 $B --len $1 --pairs $2 --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "len $1 pairs $2 48/8 streams $3 group $4" | tee -a $OUT/summary.txt
  done
done
