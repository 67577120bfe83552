#!/bin/bash
# VALU / SALU / LDS instruction counts of k_fast with phase D switched off (--debug-ablate 4) and with
# the tag probe switched off (2), against the full kernel: how the instruction budget splits.
set -e
TAG=${1:-sq_ablate}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for ab in 0 4 2; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/ab$ab" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --cpu-python-sample 0 --no-check --debug-ablate $ab > "$OUT/ab$ab.log" 2>&1
    echo "ablate $ab"; python3 "$ROOT/tools/pmc_rows.py" "$(find "$OUT/ab$ab" -name '*counter_collection.csv' | head -1)" k_fast
done
