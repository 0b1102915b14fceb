#!/bin/bash
# dynamic LDS reserved per fill workgroup (bounds the workgroups a compute unit takes) x launch shape, one call
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2; do for cfg in ${CONFIGS:-"2 2 48 8" "2 4 48 8" "4 2 48 8" "2 2 20 5"}; do :; done; done
IFS=';' read -ra CF <<< "${CONFIGS:-2 2 48 8;2 4 48 8;4 2 48 8;2 2 20 5}"
for rep in 1 2; do for cfg in "${CF[@]}"; do for pad in ${PADS:-0 24 36 40 48 56}; do set -- $cfg
  CSADP_BITS_LDS_PAD=$pad CSADP_BITS_WORDS=${W:-2} CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('rep $rep pad $pad KB streams $1 group $2 steps $3: %7.0f GCUPS %.3f ms/step  fill alone %.3f (%d passes) tb %.3f verified %s' % (d['value'], d['ms_per_step'], k['fill_launch_alone'], k['passes_in_that_launch'], k['traceback_and_expand_alone'], d.get('verified')))"
done; done; done
