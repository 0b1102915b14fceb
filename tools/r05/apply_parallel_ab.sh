#!/bin/bash
# rows and profile of a wide alignment rebuilt / compacted as items of a pool region (base) against one thread (build/libcsadp_oldhost.so): the host phases of the sets' rounds, summed
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
cp csa_amd/libcsadp.so /tmp/libcsadp_base.so
for rep in 1 2 3; do
for lib in base oldhost; do
  if [ $lib = base ]; then cp /tmp/libcsadp_base.so csa_amd/libcsadp.so; else cp build/libcsadp_$lib.so csa_amd/libcsadp.so; fi
  for s in Primates Mammals Set3; do
    CSADP_TRACE_HOST=1 timeout -k 10 200 python tools/msa_probe.py $s 2>&1 | awk -v c=$lib -v s=$s '/call 1/{p=1} /call 2/{p=0; dp=$0} p && /csadp round/{sp+=$(NF-3); cm+=$(NF-1); ap+=$(NF-6)} END{printf "%-8s %-8s: apply %.2f speculate %.2f commit %.2f ms summed over the rounds of call 2 | %s\n", c, s, ap, sp, cm, dp}'
  done
done
done
cp /tmp/libcsadp_base.so csa_amd/libcsadp.so
