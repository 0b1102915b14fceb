#!/bin/bash
# host phase timers of the drop-in program's batch (cold process), per set and mode
set -e
out=gpurun_out/r05_dropin_trace.txt
: > $out
for name in Primates Set3; do
  for bin in CSA_csadp_deferred CSA_csadp; do
    d=$(mktemp -d)
    cp tests/golden/data/$name.txt $d/
    echo "=== $bin $name" >> $out
    (cd $d && CSADP_DROPIN_TRACE=1 CSADP_TRACE_HOST=1 CSADP_DROPIN_STATS=$d/stats.json $GRAFT_REPO_ROOT/oracle/_ref/$bin $name.txt < /dev/null > $d/stdout.txt 2> $d/stderr.txt; cat $d/stderr.txt | head -400; cat $d/stats.json) >> $out 2>&1
    rm -rf $d
  done
done
