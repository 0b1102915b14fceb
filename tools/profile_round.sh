#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box:
#   tools/profile_round.sh gpurun_out/r01b      (then: python tools/summarize_profile.py gpurun_out/r01b profiles/r01)
# One --kernel-trace --stats run of bench.py, then separate --pmc passes (FETCH_SIZE and WRITE_SIZE
# do not fit one pass on gfx950, MI355X_MICROARCH.md).  The program itself follows "--".
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$ROOT/bench.py" --steps 16 --warmup 4 --no-cpu-baseline > "$OUT/log_stats.txt" 2>&1
for C in WRITE_SIZE FETCH_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
	TAG=$(echo $C | cut -d' ' -f1)
	rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o pmc_$TAG -- python3 "$ROOT/bench.py" --steps 8 --warmup 0 --no-cpu-baseline > "$OUT/log_$TAG.txt" 2>&1
done
ls "$OUT"
# launches one at a time (a single stream): the per-launch duration bench.py reports as avg_launch_us
CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o solo -- python3 "$ROOT/bench.py" --steps 16 --warmup 4 --no-cpu-baseline > "$OUT/log_solo.txt" 2>&1
