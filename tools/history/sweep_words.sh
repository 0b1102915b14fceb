#!/bin/bash
# words per lane x fill launches in flight x passes per launch, all inside one call (boxes differ by up to 5 %):
# tools/sweep_words.sh ["W streams group;..."] ["steps warmup;..."]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
CONFIGS=${1:-"1 2 2;2 2 2;2 2 4;2 4 2;2 3 4;4 2 4;4 2 8"}
RUNS=${2:-"20 5;48 8"}
IFS=';' read -ra CF <<< "$CONFIGS"
IFS=';' read -ra RN <<< "$RUNS"
for rep in 1 2; do for run in "${RN[@]}"; do for cfg in "${CF[@]}"; do set -- $run $cfg
  CSADP_BITS_WORDS=$3 CSADP_BITS_STREAMS=$4 CSADP_BITS_GROUP=$5 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('rep $rep steps $1 W $3 streams $4 group $5: %7.0f GCUPS  %.3f ms/step  fill alone %.3f ms (%d passes) traceback+expand alone %.3f  verified %s' % (d['value'], d['ms_per_step'], k['fill_launch_alone'], k['passes_in_that_launch'], k['traceback_and_expand_alone'], d.get('verified')))"
done; done; done
