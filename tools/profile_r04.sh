#!/bin/bash
# Round-4 profile set (one gpurun call).  Results: gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>/.
#   1. the default bench under rocprofv3 --kernel-trace --stats (the kernel's average duration beside the bench's own
#      HIP-event time; the bench's live counter passes are off inside a profiled run)
#   2. the default bench by itself (with its live FETCH_SIZE / WRITE_SIZE passes): the line the driver would see
#   3. SQ instruction mix and waits of the main pass (tools/sq2.sh)
#   4. k_split2's duration (tools/split_kernels.sh)
#   5. the device gzip decoder's stages and kernels (tools/gz_tier.py)
TAG=${1:-r04_final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
echo "[profile] kernel trace of the default bench"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --traffic off --tier-reads 0 --cpu-sample 0 --other-configs= > "$OUT/bench_traced.log" 2> "$OUT/bench_traced.err"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
echo "[profile] the default bench"
python3 "$ROOT/bench.py" > "$OUT/bench_default.log" 2> "$OUT/bench_default.err"
echo "[profile] SQ counters"
cd "$ROOT" && tools/sq2.sh ${TAG}_sq > /dev/null 2>&1; cp gpurun_out/${TAG}_sq/sq.txt "$OUT/sq_main_pass.txt"
echo "[profile] k_split2"
READS=40000000 tools/split_kernels.sh libtagdig.so > "$OUT/split_kernels.txt" 2>&1
echo "[profile] k_count (the exact in-flight kernel: tassel_tagcount, early maxreads, matrices of 4 GiB and more)"
python3 "$ROOT/bench.py" --opt fastpath=0 --steps 3 --warmup 1 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 --other-configs= --traffic off 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('k_count (fastpath=0): kernel_ms %.3f  frac %.3f  bit-exact %s' % (o['roofline']['kernel_ms'], o['roofline']['frac'], o['check']['bit_exact_vs_expected']))" > "$OUT/k_count.txt" 2>&1
echo "[profile] read lengths, CRLF"
python3 "$ROOT/tools/readlen_sweep.py" > "$OUT/readlen.txt" 2>&1
python3 "$ROOT/tools/crlf_check.py" 8000000 > "$OUT/crlf.txt" 2>&1
echo "[profile] ordinary gzip decoded on the device: stage times, then the kernels under rocprofv3"
TAGDIG_INFLATE_STATS=1 python3 "$ROOT/tools/gz_tier.py" > "$OUT/gzip_gpu_tier.txt" 2> "$OUT/gzip_gpu_tier.err"
grep -a "gz_gpu_inflate: segment\|count_gzip_gpu" "$OUT/gzip_gpu_tier.err" | tail -4 >> "$OUT/gzip_gpu_tier.txt"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gzstats" -- python3 "$ROOT/tools/gz_tier.py" 16000000 2 > "$OUT/gzip_gpu_traced.log" 2>&1
f=$(find "$OUT/gzstats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -a "Name\|tdgz\|k_fast\|k_resolve" "$f" > "$OUT/gzip_gpu_kernel_stats.csv"
rm -rf "$OUT/gzstats"
rm -rf "$OUT/stats"
ls -la "$OUT"
echo "[profile] k_fast2 against k_fast4 (the default) and against round 3's kernel, passes alternating in one process"
cd "$ROOT" && python3 tools/ab_inproc.py libtagdig_r3.so libtagdig.so libtagdig.so --arm "" --arm kernel=2 --arm kernel=4 --reads 100000000 --rounds 10 > "$OUT/ab_r3_fast2_fast4.txt" 2>&1
ls -la "$OUT"
