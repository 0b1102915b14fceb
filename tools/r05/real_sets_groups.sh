#!/bin/bash
# real pair sets: FEWER, LARGER launches whose waves in flight fall just short of a whole number per SIMD (66 jobs x 3 waves x 10 passes = 1.93,
# x 15 = 2.9; 120 x 3 x 14 = 4.92, x 8 = 2.81, x 11 = 3.87)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --mode strong --workload $1 --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $TAG: %.1f TCUPS  %.3f ms/step  passes/launch %s streams %s verified %s' % (d['value']/1e3, d['ms_per_step'], d['config']['passes_per_launch'], d['config']['launches_in_flight'], d['verified']))"
}
for rep in 1 2; do
for wl in mammals primates; do
  TAG="default"; run $wl
  for cfg in "2 5" "3 5" "2 7" "2 4" "4 4" "1 8" "2 8" "3 4"; do set -- $cfg
    TAG="streams $1 group $2"; CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 run $wl
  done
done
done
