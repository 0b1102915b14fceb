#!/bin/bash
# The bench part of tools/profile_round.sh alone: kernel stats (pipelined and one launch at a time) and the PMC passes.
#   tools/profile_bench.sh gpurun_out/r02x
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
PMC=("WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_stats -- $BENCH --steps 16 --warmup 4 > "$OUT/log_bench_stats.txt" 2>&1
CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_solo -- $BENCH --steps 16 --warmup 4 > "$OUT/log_bench_solo.txt" 2>&1
for C in "${PMC[@]}"; do
	TAG=$(echo $C | cut -d' ' -f1)
	CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o bench_pmc_$TAG -- $BENCH --steps 8 --warmup 0 > "$OUT/log_bench_$TAG.txt" 2>&1 || echo "pmc pass $TAG failed"
done
ls "$OUT"
