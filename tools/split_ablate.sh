#!/bin/bash
# k_split2's kernel time with parts switched off (timing only): TD_ABLATE 0 / 64 (no site search) / 128 (no adapter search) / 192
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for a in ${1:-0 64 128 192}; do
  out=$ROOT/gpurun_out/splita_$a
  TD_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $ROOT/tools/split_bench.py ${2:-40000000} 2 nofile > $out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep "k_split2\|k_count_lines<6>\|k_scan" "$f" | cut -d, -f1-4 | cut -c1-100 | sed "s/^/ablate $a: /"
done
