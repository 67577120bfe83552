#!/bin/bash
# The short profile set for the round's last commits (one gpurun call, about five minutes): the default bench by itself and
# under rocprofv3 --kernel-trace --stats, and the device gzip decoder's stages and kernels.  tools/profile_r04.sh is the long one.
TAG=${1:-r04_head}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
echo "[profile] the default bench"
python3 "$ROOT/bench.py" > "$OUT/bench_default.log" 2> "$OUT/bench_default.err" || exit 1
echo "[profile] kernel trace of the default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --traffic off --tier-reads 0 --cpu-sample 0 --other-configs= > "$OUT/bench_traced.log" 2> "$OUT/bench_traced.err" || exit 1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
echo "[profile] ordinary gzip decoded on the device: stage times, then the kernels under rocprofv3"
TAGDIG_INFLATE_STATS=1 python3 "$ROOT/tools/gz_tier.py" > "$OUT/gzip_gpu_tier.txt" 2> "$OUT/gzip_gpu_tier.err" || exit 1
grep -a "gz_gpu_inflate: segment\|count_gzip_gpu" "$OUT/gzip_gpu_tier.err" | tail -4 >> "$OUT/gzip_gpu_tier.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gzstats" -- python3 "$ROOT/tools/gz_tier.py" 16000000 2 > "$OUT/gzip_gpu_traced.log" 2>&1 || exit 1
f=$(find "$OUT/gzstats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -a "Name\|tdgz\|k_fast\|k_resolve" "$f" > "$OUT/gzip_gpu_kernel_stats.csv"
rm -rf "$OUT/gzstats"
ls -la "$OUT"
