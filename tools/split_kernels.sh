#!/bin/bash
# k_split2's duration (rocprofv3 kernel trace) for several builds of the library: tools/split_kernels.sh lib1.so lib2.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  out=$ROOT/gpurun_out/splitk_$(basename $lib .so)
  TAGDIG_LIB=$ROOT/tagdigger_amd/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 $ROOT/tools/split_bench.py ${READS:-40000000} 2 nofile > $out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep "k_split2\|k_count_lines<6>\|k_scan" "$f" | cut -d, -f1-4 | cut -c1-110 | sed "s/^/$lib: /"
done
