#!/bin/bash
# A/B of library builds inside one call: tools/ab_lib.sh build/libcsadp_X.so ...  (the shipped library is "base");
# CONFIGS="streams group steps warmup;..." overrides the launch shapes tried
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
CONFIGS=${CONFIGS:-"2 2 48 8;4 2 48 8;2 4 48 8"}
run() {
  IFS=';' read -ra CF <<< "$CONFIGS"
  for cfg in "${CF[@]}"; do set -- $cfg
    CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$TAG streams $1 group $2 steps $3: %.3f ms/step  alone %.3f (%d passes) verified %s' % (d['ms_per_step'], d['kernel_ms']['fill_launch_alone'], d['kernel_ms']['passes_in_that_launch'], d.get('verified')))"
  done
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  for lib in "$@"; do TAG=$(basename $lib .so); cp $lib csa_amd/libcsadp.so; run; done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
