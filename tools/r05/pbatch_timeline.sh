#!/bin/bash
# kernel timeline of one N-sequence throughput batch (which kernels run at once, how long the device idles between rounds)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SHAPE=${1:-64x4x30000}
OUT=$ROOT/gpurun_out/pbtl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python3 $ROOT/tools/r05/profile_batch_probe.py $SHAPE > $OUT/log.txt 2>&1
grep "call" $OUT/log.txt | cut -c1-200
F=$(find $OUT -name "tl_kernel_trace.csv" | head -1)
python3 $ROOT/tools/r05/timeline.py $F
