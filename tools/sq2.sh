#!/bin/bash
# SQ counter passes (separate rocprofv3 --pmc runs of a short bench) for whatever kernel the bench arguments
# select.  usage: tools/sq2.sh <tag> [bench args...]   -> gpurun_out/<tag>/sq.txt
set -e
TAG=${1:-sq}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 --no-check --other-configs= --traffic off $*"
: > "$OUT/sq.txt"
echo "# bench args: $ARGS" >> "$OUT/sq.txt"
pass() {   # name, counters...
    local name=$1; shift
    if rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1; then
        f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && python3 "$ROOT/tools/pmc_rows.py" "$f" k_fast >> "$OUT/sq.txt"
    else
        echo "  (pass $name failed: see $OUT/$name.log)" >> "$OUT/sq.txt"; tail -3 "$OUT/$name.log"
    fi
    rm -rf "$OUT/$name"
}
pass mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
cat "$OUT/sq.txt"
