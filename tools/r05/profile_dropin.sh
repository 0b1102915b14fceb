#!/bin/bash
# rocprofv3 kernel statistics of the reference PROGRAM relinked with the drop-in (deferred mode), mode N on Primates and Set3
cd /tmp && export TMPDIR=/tmp
for name in Primates Set3; do
  d=$(mktemp -d); cp $GRAFT_REPO_ROOT/tests/golden/data/$name.txt $d/; cd $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_dropin_prof -o $name -- $GRAFT_REPO_ROOT/oracle/_ref/CSA_csadp_deferred $name.txt < /dev/null > stdout.txt 2> stderr.txt
  md5sum $name-Aligned.fasta
  cd /tmp; rm -rf $d
done
find $GRAFT_REPO_ROOT/gpurun_out/r05_dropin_prof -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -12 $f | cut -c1-160; done
