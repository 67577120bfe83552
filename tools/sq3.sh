#!/bin/bash
# instruction mix of the main pass per ablation mask (one rocprofv3 --pmc pass each)  usage: tools/sq3.sh <tag> "<masks>" [bench args]
TAG=${1:-sq3}; MASKS=${2:-"0 4"}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
: > "$OUT/sq3.txt"
for m in $MASKS; do
  echo "== ablate $m $*" >> "$OUT/sq3.txt"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/p$m" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 --no-check --debug-ablate $m "$@" > "$OUT/p$m.log" 2>&1
  f=$(find "$OUT/p$m" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 "$ROOT/tools/pmc_rows.py" "$f" k_fast >> "$OUT/sq3.txt"
  rm -rf "$OUT/p$m"
done
cat "$OUT/sq3.txt"
