#!/bin/bash
# config 4: launches in flight x passes per launch x LDS reservation, now that a traceback workgroup takes 92 KB (round 3: 113 KB at two words per lane)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1
mkdir -p "$OUT"
cd $ROOT
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
fmt='import json,sys; d=json.loads(sys.stdin.read()); print("%s: %.0f GCUPS  %.3f ms/step  verified %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("verified")))'
for cfg in "2 2 27" "4 2 10" "4 2 12" "3 2 16" "3 2 12" "4 1 10" "4 1 27" "2 4 27" "3 1 16" "4 2 0" "3 2 0"; do set -- $cfg
  for st in "20 5" "48 8"; do set -- $cfg $st
    CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 CSADP_BITS_LDS_PAD=$3 $B --steps $4 --warmup $5 2>/dev/null | python3 -c "$fmt" "config4 $4/$5 streams $1 group $2 pad $3" | tee -a $OUT/summary.txt
  done
done
