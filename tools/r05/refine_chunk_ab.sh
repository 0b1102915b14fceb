#!/bin/bash
# (needs a build in which that size is read from CSADP_REFINE_CHUNK -- a four-line patch of csadp_progressive.cpp / csadp_config.h; the shipped code has 256 as a constant)
# columns per speculation item of DeleteGappedColumns (CSADP_REFINE_CHUNK): the refine phases of the example sets' rounds, summed
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for c in 256 128 64 32; do
  for s in Primates Mammals Set3; do
    CSADP_REFINE_CHUNK=$c CSADP_TRACE_HOST=1 timeout -k 10 200 python tools/msa_probe.py $s 2>&1 | awk -v c=$c -v s=$s '/call 1/{p=1} /call 2/{p=0; dp=$0} p && /csadp round/{sp+=$(NF-3); cm+=$(NF-1); ap+=$(NF-6)} END{printf "chunk %3d %-8s: speculate %.2f commit %.2f apply %.2f ms summed over the rounds of call 2 | %s\n", c, s, sp, cm, ap, dp}'
  done
done
done
