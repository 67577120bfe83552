#!/bin/bash
# A/B of library variants on one GPU box: interleaved bench runs (run-to-run noise is +-2 %).
#   usage: tools/ab_bench.sh <rounds> <variant> [<variant> ...]     ("" = libtagdig.so, "x" = libtagdig_x.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=$ROOT/tagdigger_amd/libtagdig${v:+_$v}.so
    TAGDIG_LIB=$lib timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --cpu-python-sample 0 2>&1 | tail -1 | python3 -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('%-8s ms_per_step %.3f  kernel_ms %s  frac %.4f' % ('${v:-main}', d['ms_per_step'], d['roofline'].get('kernel_ms'), d['roofline']['frac']))
except Exception as e: print('${v:-main}', 'FAILED', l[-300:])
"
  done
done
