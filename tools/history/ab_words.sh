#!/bin/bash
# A/B of library builds at fixed words per lane, all in one call: W=2 tools/ab_words.sh build/libcsadp_X.so ...
# CONFIGS="streams group steps warmup;..."
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
CONFIGS=${CONFIGS:-"2 2 20 5;2 2 48 8"}
W=${W:-2}
run() {
  IFS=';' read -ra CF <<< "$CONFIGS"
  for cfg in "${CF[@]}"; do set -- $cfg
    CSADP_BITS_WORDS=$W CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('$TAG W $W streams $1 group $2 steps $3: %7.0f GCUPS %.3f ms/step  fill alone %.3f (%d passes) tb %.3f verified %s' % (d['value'], d['ms_per_step'], k['fill_launch_alone'], k['passes_in_that_launch'], k['traceback_and_expand_alone'], d.get('verified')))"
  done
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  for lib in "$@"; do TAG=$(basename $lib .so); cp $lib csa_amd/libcsadp.so; run; done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
