#!/bin/bash
# Round-2 profile set (one gpurun call): kernel trace + FETCH/WRITE passes of the default bench, SQ instruction mix and
# waits, the same for k_fast (round 1's main pass) for comparison, and WRITE_SIZE per counted read on the skewed stream
# with and without the hot-cell cache.  Results: gpurun_out/<tag>/ -> profiles/<tag>/ (tools/pmc_summary.py + copies below)
TAG=${1:-r02_final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
tools/profile_round.sh $TAG --tier-reads 0 --oracle-sample 0 > "$OUT/profile_round.log" 2>&1
python3 tools/pmc_summary.py $TAG > "$OUT/pmc_summary.log" 2>&1
tools/sq2.sh ${TAG}_sq > /dev/null 2>&1; cp gpurun_out/${TAG}_sq/sq.txt "$OUT/sq_k_fast2.txt"
tools/sq2.sh ${TAG}_sq1 --opt kernel=1 > /dev/null 2>&1; cp gpurun_out/${TAG}_sq1/sq.txt "$OUT/sq_k_fast.txt"
cd /tmp
for hc in 1 0; do
  for sk in 0 1.5; do
    extra=""; [ "$sk" != "0" ] && extra="--skew $sk"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w_${hc}_${sk}" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 --no-check --opt hot_cache=$hc $extra > "$OUT/w_${hc}_${sk}.log" 2>&1
    f=$(find "$OUT/w_${hc}_${sk}" -name "*counter_collection.csv" | head -1)
    echo "hot_cache=$hc skew=$sk: $(python3 "$ROOT/tools/pmc_rows.py" "$f" k_fast2 | grep WRITE_SIZE)  kernel_ms $(grep -o '"kernel_ms": [0-9.]*' "$OUT/w_${hc}_${sk}.log" | head -1)" >> "$OUT/write_per_hit.txt"
    rm -rf "$OUT/w_${hc}_${sk}"
  done
done
cat "$OUT/write_per_hit.txt"
