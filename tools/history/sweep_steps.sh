#!/bin/bash
# the bench's value at the driver's arguments (20 timed steps after 5) and sustained (48 after 8), three times each
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2 3; do for kw in "20 5" "48 8"; do set -- $kw
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps $1: %.0f GCUPS  %.3f ms/step  alone %.3f' % (d['value'], d['ms_per_step'], d['kernel_ms']['fill_launch_alone']))"
done; done
