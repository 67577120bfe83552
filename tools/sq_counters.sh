#!/bin/bash
# Where k_fast's cycles go, from the SQ counters (separate rocprofv3 --pmc passes of a short bench):
# instruction mix, busy / wait cycles, LDS bank conflicts.  Output: gpurun_out/<tag>/sq_*.csv
#   usage: tools/sq_counters.sh <tag>
set -e
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --cpu-python-sample 0 --no-check"
pass() {   # name, counters...
    local name=$1; shift
    echo "[sq] $name: $*"
    if rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1; then
        f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && python3 "$ROOT/tools/pmc_rows.py" "$f" k_fast > "$OUT/sq_$name.txt" && cat "$OUT/sq_$name.txt"
    else
        echo "  (pass failed: see $OUT/$name.log)"; tail -3 "$OUT/$name.log"
    fi
}
pass mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
pass derived VALUBusy LDSBankConflict MemUnitStalled GPUBusy
