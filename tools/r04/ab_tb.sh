#!/bin/bash
# A/B of library builds inside one call on the three pair workloads: tools/r04/ab_tb.sh <tag> build/libcsadp_X.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  tb alone %.3f  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["kernel_ms"]["traceback_and_expand_alone"], d.get("verified")))'
run() {
  $B --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "$TAG config4 20/5" | tee -a $OUT/summary.txt
  for w in mammals primates; do
    $B --mode strong --workload $w --steps 20 --warmup 5 2>/dev/null | python3 -c "$fmt" "$TAG $w 20/5" | tee -a $OUT/summary.txt
    $B --mode strong --workload $w --steps 48 --warmup 8 2>/dev/null | python3 -c "$fmt" "$TAG $w 48/8" | tee -a $OUT/summary.txt
  done
}
for rep in 1 2; do
  TAG=base; cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; run
  for lib in "$@"; do TAG=$(basename $lib .so | sed s/libcsadp_//); cp $lib csa_amd/libcsadp.so; run; done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
