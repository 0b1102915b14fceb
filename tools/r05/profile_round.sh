#!/bin/bash
# Round 5's rocprofv3 evidence (tools/profile_round.sh + the N-sequence batch workload):
#   tools/r05/profile_round.sh gpurun_out/r05prof   (then: python tools/summarize_profile.py gpurun_out/r05prof profiles/r05)
#   bench    bench.py's timed region (nw_pack_planes, nw_fill_bits, nw_traceback_windows, nw_expand_rows), pipelined and one launch at a time
#   msa      tools/msa_probe.py (mode N of the example sets): nw_fill_cells latency-shaped, nw_tb_*
#   pbatch   tools/r05/profile_batch_probe.py 256x8x4000: nw_fill_cells in launches of 2 304 workgroups (two to a compute unit)
# Counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md); the program itself follows "--".
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-legs"
PBATCH="python3 $ROOT/tools/r05/profile_batch_probe.py 256x8x4000"
# ONLY="msa pbatch" tools/r05/profile_round.sh DIR  re-runs those workloads only (the others' files in DIR stay as they are)
want() { [[ -z "$ONLY" || " $ONLY " == *" $1 "* ]]; }
PMC=("WRITE_SIZE" "FETCH_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS")
if want bench; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_stats -- $BENCH --steps 16 --warmup 4 > "$OUT/log_bench_stats.txt" 2>&1
echo bench stats done
CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench_solo -- $BENCH --steps 16 --warmup 4 > "$OUT/log_bench_solo.txt" 2>&1
for C in "${PMC[@]}"; do
	TAG=$(echo $C | cut -d' ' -f1)
	CSADP_BITS_STREAMS=1 rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o bench_pmc_$TAG -- $BENCH --steps 8 --warmup 0 > "$OUT/log_bench_$TAG.txt" 2>&1
done
echo bench pmc done
fi
if want msa; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o msa_stats -- python3 $ROOT/tools/msa_probe.py > "$OUT/log_msa_stats.txt" 2>&1
for C in "${PMC[@]}"; do
	TAG=$(echo $C | cut -d' ' -f1)
	rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o msa_pmc_$TAG -- python3 $ROOT/tools/msa_probe.py Set3 > "$OUT/log_msa_$TAG.txt" 2>&1
done
echo msa done
fi
if want pbatch; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o pbatch_stats -- $PBATCH > "$OUT/log_pbatch_stats.txt" 2>&1
CSADP_CELLS_FETCH=100000 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o pbatchfetch_stats -- $PBATCH > "$OUT/log_pbatchfetch_stats.txt" 2>&1
for C in "${PMC[@]:2}"; do
	TAG=$(echo $C | cut -d' ' -f1)
	rocprofv3 --kernel-trace --output-format csv --pmc $C -d "$OUT" -o pbatch_pmc_$TAG -- $PBATCH > "$OUT/log_pbatch_$TAG.txt" 2>&1
done
echo pbatch done
fi
ls "$OUT" | head -80
