#!/bin/bash
# HIP API trace of the drop-in program (deferred mode): which calls of the first real batch are slow?
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d); cp $GRAFT_REPO_ROOT/tests/golden/data/Primates.txt $d/
cd $d
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_dropin_hiptrace -- $GRAFT_REPO_ROOT/oracle/_ref/CSA_csadp_deferred Primates.txt < /dev/null > stdout.txt 2> stderr.txt
tail -3 stderr.txt
cd $GRAFT_REPO_ROOT/gpurun_out/r05_dropin_hiptrace && find . -name "*.csv" | head; 
f=$(find . -name "*hip_api_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
big=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]),r["Function"],int(r["Start_Timestamp"])) for r in rows]
t0=min(b[2] for b in big)
for d,f,s in sorted(big,reverse=True)[:40]:
    print("%9.3f ms  at %9.3f ms  %s"%(d/1e6,(s-t0)/1e6,f))
PY
