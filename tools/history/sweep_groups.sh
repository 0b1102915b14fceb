#!/bin/bash
# merged passes per launch x launches in flight: which combination sustains the highest rate on the bench batch
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2; do
for s in 2 3 4; do for g in 1 2 3 4 8; do
  CSADP_SLOTS=4 CSADP_BITS_GROUP=$g CSADP_BITS_STREAMS=$s python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rep $rep streams $s group $g: %.0f GCUPS  %.3f ms/step  alone %.3f ms for %d passes' % (d['value'], d['ms_per_step'], d['kernel_ms']['fill_launch_alone'], d['kernel_ms']['passes_in_that_launch']))"
done; done; done
