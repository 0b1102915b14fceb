#!/bin/bash
# config 4 (128 pairs of 16 kbp, two words per lane): LDS reservation of the fill workgroups now that a traceback workgroup takes 92 KB
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd $ROOT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("verified")))'
for rep in 1 2; do
for pad in default 0 16 24 32 40 48 56; do
  for st in "20 5" "48 8"; do set -- $st
    if [ $pad = default ]; then $B --steps $1 --warmup $2 2>/dev/null | python3 -c "$fmt" "config4 $1/$2 pad default" | tee -a $OUT/summary.txt
    else CSADP_BITS_LDS_PAD=$pad $B --steps $1 --warmup $2 2>/dev/null | python3 -c "$fmt" "config4 $1/$2 pad $pad" | tee -a $OUT/summary.txt; fi
  done
done
done
