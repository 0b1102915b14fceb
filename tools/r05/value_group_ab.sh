#!/bin/bash
# `value` (128 pairs of 16 kbp, the driver's 5 + 20 steps, and 8 + 48) under passes per launch 2 (shipped) / 4 / 6 / 8 with two launches in flight
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for g in default 4 6 8; do
  for sw in "20 5" "48 8"; do set -- $sw
    if [ $g = default ]; then E=""; else E="CSADP_BITS_GROUP=$g"; fi
    env $E python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('group $g steps $1: %.1f TCUPS (from idle %.1f)  %.3f ms/step  passes/launch %s streams %s verified %s' % (d['value']/1e3, d['config']['value_from_idle_gcups']/1e3, d['ms_per_step'], d['config']['passes_per_launch'], d['config']['launches_in_flight'], d['verified']))"
  done
done
done
