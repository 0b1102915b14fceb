#!/bin/bash
# Throughput of pair batches whose shapes do not fill the chip evenly (GPU box): pairs x length, then the real sets
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "128 16384" "120 16384" "120 17000" "100 16384" "66 17000" "200 16384" "256 8192" "64 33000"; do
  set -- $cfg
  python bench.py --pairs $1 --len $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 pairs x $2: %6.0f GCUPS  %.3f ms/step  W %d  passes/launch %d  verified %s' % (d['value'], d['ms_per_step'], d['config']['words_per_lane'], d['config']['passes_per_launch'], d.get('verified')))"
done
for w in primates mammals; do
  python bench.py --mode strong --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w: %6.0f GCUPS  %.3f ms/step  W %d  passes/launch %d  verified %s' % (d['value'], d['ms_per_step'], d['config']['words_per_lane'], d['config']['passes_per_launch'], d.get('verified')))"
done
