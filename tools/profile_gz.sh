#!/bin/bash
# The device gzip decoder by itself (one gpurun call): its tests, the 16 M-read tier's stage times, the kernels under
# rocprofv3 --kernel-trace --stats, and (E2E=1) BASELINE configs[2] as written -- one 200 M-read gzip member end to end.
TAG=${1:-gz}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
echo "[gz] tests"
timeout -k 10 500 python3 -m pytest tests/test_gzip_gpu_huffman.py tests/test_gzip_gpu.py tests/test_gzip_damage.py -m gpu -x -q > "$OUT/gzip_tests.txt" 2>&1 || { tail -20 "$OUT/gzip_tests.txt"; exit 1; }
tail -2 "$OUT/gzip_tests.txt"
cd /tmp
echo "[gz] stage times"
TAGDIG_INFLATE_STATS=1 python3 "$ROOT/tools/gz_tier.py" > "$OUT/gzip_gpu_tier.txt" 2> "$OUT/gzip_gpu_tier.err" || exit 1
grep -a "gz_gpu_inflate: segment\|count_gzip_gpu" "$OUT/gzip_gpu_tier.err" | tail -4 >> "$OUT/gzip_gpu_tier.txt"
cat "$OUT/gzip_gpu_tier.txt"
echo "[gz] kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gzstats" -- python3 "$ROOT/tools/gz_tier.py" 16000000 2 > "$OUT/gzip_gpu_traced.log" 2>&1 || exit 1
f=$(find "$OUT/gzstats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -a "Name\|tdgz\|k_fast\|k_resolve" "$f" > "$OUT/gzip_gpu_kernel_stats.csv"
rm -rf "$OUT/gzstats"
if [ "${E2E:-0}" = 1 ]; then
  echo "[gz] configs[2] as written"
  TAGDIG_INFLATE_STATS=1 python3 "$ROOT/tools/config3_gzip_e2e.py" > "$OUT/config3_gzip_file_200M_reads_device_decoder.txt" 2>&1 || { tail -5 "$OUT/config3_gzip_file_200M_reads_device_decoder.txt"; exit 1; }
  grep -a "call\|count_gzip_gpu" "$OUT/config3_gzip_file_200M_reads_device_decoder.txt"
fi
