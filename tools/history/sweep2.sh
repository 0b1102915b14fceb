#!/bin/bash
# passes per launch x fill launches in flight (each with its traceback on a side stream): sustained ms per pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2; do for kw in "20 5" "48 8"; do set -- $kw; for sg in "1 4" "1 8" "2 2" "2 3" "2 4" "2 6" "2 8" "3 2"; do set -- $kw $sg
  CSADP_BITS_GROUP=$4 CSADP_BITS_STREAMS=$3 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rep $rep steps $1 streams $3 group $4: %.0f GCUPS  %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done; done
