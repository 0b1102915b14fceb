#!/bin/bash
# streams x passes per launch, sustained ms per pass (48 timed steps) -- after the carry-mask step (more waves per SIMD pay)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for sg in "2 2" "2 4" "3 2" "3 4" "4 2" "4 4" "2 8" "1 4"; do set -- $sg
  CSADP_BITS_GROUP=$2 CSADP_BITS_STREAMS=$1 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 group $2: %.0f GCUPS  %.3f ms/step' % (d['value'], d['ms_per_step']))"
done
