#!/bin/bash
out=gpurun_out/r05_dropin_variants.txt
: > $out
run() {
  d=$(mktemp -d); cp tests/golden/data/Primates.txt $d/
  echo "=== $*" >> $out
  (cd $d && env "$@" CSADP_DROPIN_TRACE=1 CSADP_DROPIN_STATS=$d/stats.json $GRAFT_REPO_ROOT/oracle/_ref/CSA_csadp_deferred Primates.txt < /dev/null > $d/stdout.txt 2> $d/stderr.txt; grep -B70 "call 1 " $d/stderr.txt | grep -v "^csadp round\|csadp_pairs" ; cat $d/stats.json) >> $out 2>&1
  rm -rf $d
}
run CSADP_WARMUP_PARTS=0
run CSADP_DROPIN_EARLY_INIT=0 CSADP_WARMUP_PARTS=0
