#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box:
#   tools/profile_round.sh gpurun_out/r03   (then: python tools/summarize_profile.py gpurun_out/r03 profiles/r03)
# Per workload one --kernel-trace --stats run, then separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950, MI355X_MICROARCH.md).  The program itself follows "--".
#   bench    bench.py's timed region: nw_pack_planes, nw_fill_bits, nw_traceback_windows, nw_expand_rows
#   msa      tools/msa_probe.py (mode N of the example sets): nw_fill_cells, nw_tb_scout / _resolve / _emit / _gather
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
PMC=("WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_stats -- $BENCH --steps 16 --warmup 4 > "$OUT/log_bench_stats.txt" 2>&1
echo bench stats done
# launches one at a time (a single stream): the per-launch duration bench.py reports as avg_launch_us
CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_solo -- $BENCH --steps 16 --warmup 4 > "$OUT/log_bench_solo.txt" 2>&1
for C in "${PMC[@]}"; do
	TAG=$(echo $C | cut -d' ' -f1)
	CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o bench_pmc_$TAG -- $BENCH --steps 8 --warmup 0 > "$OUT/log_bench_$TAG.txt" 2>&1
done
echo bench pmc done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o msa_stats -- python3 $ROOT/tools/msa_probe.py > "$OUT/log_msa_stats.txt" 2>&1
for C in "${PMC[@]}"; do
	TAG=$(echo $C | cut -d' ' -f1)
	rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o msa_pmc_$TAG -- python3 $ROOT/tools/msa_probe.py Set3 > "$OUT/log_msa_$TAG.txt" 2>&1
done
echo msa done
ls "$OUT" | head -80
