#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2 3; do for cfg in "0 2 2" "1 2 2" "1 3 2" "1 4 2" "1 2 4" "1 4 1"; do set -- $cfg
 for kw in "48 8" "20 5"; do set -- $cfg $kw
  CSADP_BITS_CARRY=$1 CSADP_BITS_STREAMS=$2 CSADP_BITS_GROUP=$3 python bench.py --steps $4 --warmup $5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('carry $1 streams $2 group $3 steps $4: %.0f GCUPS  %.3f ms/step' % (d['value'], d['ms_per_step']))"
 done
done; done
