#!/bin/bash
# passes per launch for THREE-strip workgroups: the real sets and synthetic batches of the same shape
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --mode strong --workload $1 --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $TAG: %.1f TCUPS  passes/launch %s streams %s verified %s' % (d['value']/1e3, d['config']['passes_per_launch'], d['config']['launches_in_flight'], d['verified']))"
}
for rep in 1 2; do
for wl in mammals primates; do
  TAG="default"; run $wl
  for g in 2 3 4 5; do TAG="group $g"; CSADP_BITS_GROUP=$g run $wl; done
done
done
