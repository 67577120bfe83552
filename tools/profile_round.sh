#!/bin/bash
# Profile the headline bench on the GPU box: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE
# in separate counter passes (TCC slots do not fit both). Summaries land in gpurun_out/<tag>/;
# copy what is to be judged into profiles/<tag>/ afterwards (tools/pmc_summary.py does both).
#   usage: tools/profile_round.sh <tag> [bench args...]
set -e
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 1 --cpu-sample 0 $*"
echo "[profile] kernel-trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_stats.log" 2>&1
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS --no-check > "$OUT/bench_fetch.log" 2>&1
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS --no-check > "$OUT/bench_write.log" 2>&1
find "$OUT" -name "*.csv" | sort
