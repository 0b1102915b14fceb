#!/bin/bash
# real pair sets with more fill launches in flight than the engine's four main streams allow by default (CSADP_SLOTS raises them)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --mode strong --workload $1 --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $TAG: %.1f TCUPS  %.3f ms/step  passes/launch %s streams %s verified %s' % (d['value']/1e3, d['ms_per_step'], d['config']['passes_per_launch'], d['config']['launches_in_flight'], d['verified']))"
}
for wl in mammals primates; do
  TAG="default"; run $wl
  for cfg in "8 6 2" "8 6 3" "8 8 2" "8 8 3" "6 6 3" "8 5 3"; do set -- $cfg
    TAG="slots $1 streams $2 group $3"; CSADP_SLOTS=$1 CSADP_BITS_STREAMS=$2 CSADP_BITS_GROUP=$3 GPU_MAX_HW_QUEUES=16 run $wl
  done
done
