#!/bin/bash
# One gpurun call of a round: the -m gpu tests, the bench line, a 2-rank rehearsal of the self-launch,
# the alignment-stage probe and a rocprofv3 kernel trace of it.  usage: tools/gpu_round.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02a}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q > "$OUT/${TAG}_pytest.log" 2>&1
rc=$?
tail -5 "$OUT/${TAG}_pytest.log"
[ $rc -lt 2 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || exit $?
tail -c 600 "$OUT/${TAG}_bench.json"
timeout -k 10 300 python bench.py --gpus 2 --share-device --backend gloo --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench2.json" 2> "$OUT/${TAG}_bench2.err" || { tail -20 "$OUT/${TAG}_bench2.err"; exit 9; }
timeout -k 10 300 python tools/msa_probe.py > "$OUT/${TAG}_msa_probe.log" 2>&1 || { tail -20 "$OUT/${TAG}_msa_probe.log"; exit 8; }
cat "$OUT/${TAG}_msa_probe.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_msa_prof" -o stats -- python3 "$ROOT/tools/msa_probe.py" Primates Set3 > "$OUT/${TAG}_msa_prof.log" 2>&1 || { tail -20 "$OUT/${TAG}_msa_prof.log"; exit 7; }
find "$OUT/${TAG}_msa_prof" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
