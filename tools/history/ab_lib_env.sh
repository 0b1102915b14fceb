#!/bin/bash
# like ab_lib.sh with extra environment for every run: ENVV="CSADP_BITS_CARRY=1" tools/ab_lib_env.sh libs...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
CONFIGS=${CONFIGS:-"4 2 48 8;4 2 20 5"}
run() {
  IFS=';' read -ra CF <<< "$CONFIGS"
  for cfg in "${CF[@]}"; do set -- $cfg
    env $ENVV CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$TAG [$ENVV] streams $1 group $2 steps $3: %.3f ms/step  alone %.3f verified %s' % (d['ms_per_step'], d['kernel_ms']['fill_launch_alone'], d.get('verified')))"
  done
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  for lib in "$@"; do TAG=$(basename $lib .so); cp $lib csa_amd/libcsadp.so; run; done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
