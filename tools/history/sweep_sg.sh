#!/bin/bash
# streams x passes per launch (group), sustained ms per pass over 48 timed steps, two repetitions
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2; do for sg in "2 2" "3 2" "4 2" "2 4" "2 3" "3 1" "4 1"; do set -- $sg
  CSADP_BITS_GROUP=$2 CSADP_BITS_STREAMS=$1 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 group $2: %.0f GCUPS  %.3f ms/step  alone %.3f' % (d['value'], d['ms_per_step'], d['kernel_ms']['fill_launch_alone']))"
done; done
