#!/bin/bash
# SQ counter passes (separate rocprofv3 --pmc runs) over the device gzip decoder's kernels on the tiers' 16 M-read file.
# usage: tools/sq_gz.sh <tag>   -> gpurun_out/<tag>/sq_gz.txt
TAG=${1:-sq_gz}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
: > "$OUT/sq_gz.txt"
pass() {   # name, counters...
    local name=$1; shift
    if rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/gz_tier.py" 16000000 1 > "$OUT/$name.log" 2>&1; then
        f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
        for k in k_gz_lz k_gz_tokens k_gz_find k_gz_crc; do
            echo "-- $k" >> "$OUT/sq_gz.txt"
            [ -n "$f" ] && python3 "$ROOT/tools/pmc_rows.py" "$f" $k >> "$OUT/sq_gz.txt"
        done
    else
        echo "  (pass $name failed: see $OUT/$name.log)" >> "$OUT/sq_gz.txt"; tail -3 "$OUT/$name.log"
    fi
    rm -rf "$OUT/$name"
}
pass mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU
cat "$OUT/sq_gz.txt"
