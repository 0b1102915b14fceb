#!/bin/bash
# (needs a build with nw_fill_bits<W, 4, true, 4> and its work list: not in the shipped code -- see profiles/r05_real_sets_probes.txt)
# three-strip jobs (the real pair sets at three words per lane) as chains of workgroups that hold the same strip of four jobs (CSADP_BITS_PACK=2) against one three-wave workgroup per job
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --mode strong --workload $1 --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 $TAG: %.1f TCUPS  %.3f ms/step  passes/launch %s streams %s verified %s' % (d['value']/1e3, d['ms_per_step'], d['config']['passes_per_launch'], d['config']['launches_in_flight'], d['verified']))"
}
for rep in 1 2; do
for wl in mammals primates; do
  TAG="default"; run $wl
  TAG="transposed"; CSADP_BITS_PACK=2 run $wl
  for cfg in "2 2" "2 4" "2 5" "2 8" "3 4" "4 2" "4 4" "3 5"; do set -- $cfg
    TAG="transposed streams $1 group $2"; CSADP_BITS_PACK=2 CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 run $wl
  done
done
done
