#!/bin/bash
# config 4 (128 pairs of 16 384) at FOUR words per lane (two strips per job: 94 VALU per 128 cells against 52 per 64) in several launch shapes, against the default (two words, 2 x 2)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$TAG: %.1f TCUPS  %.3f ms/step  words %s passes/launch %s streams %s verified %s' % (d['value']/1e3, d['ms_per_step'], d['config'].get('words_per_lane'), d['config'].get('passes_per_launch'), d['config'].get('launches_in_flight'), d.get('verified')))"
}
for rep in 1 2; do
TAG="default"; run
for cfg in "4 -1 -1" "4 2 4" "4 4 4" "4 2 8" "4 4 2" "4 3 4" "4 2 6" "3 -1 -1"; do set -- $cfg
  TAG="words $1 streams $2 group $3"
  if [ $2 = -1 ]; then CSADP_BITS_WORDS=$1 run; else CSADP_BITS_WORDS=$1 CSADP_BITS_STREAMS=$2 CSADP_BITS_GROUP=$3 run; fi
done
done
