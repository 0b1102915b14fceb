#!/bin/bash
# A/B of one environment switch inside one call: tools/ab.sh VAR A B [steps warmup]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
VAR=$1; A=$2; B=$3; K=${4:-48}; W=${5:-8}
for rep in 1 2 3; do for v in $A $B; do
  env $VAR=$v python bench.py --steps $K --warmup $W --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v steps $K: %.0f GCUPS  %.3f ms/step  alone %.3f tb %.3f verified %s' % (d['value'], d['ms_per_step'], d['kernel_ms']['fill_launch_alone'], d['kernel_ms']['traceback_and_expand_alone'], d.get('verified')))"
done; done
