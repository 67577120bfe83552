#!/bin/bash
# k_fast wave priorities (bench.py --prio: bits 1:0 phase A, 3:2 B-C, 5:4 D, 7:6 A-end), interleaved rounds
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    python3 $ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --cpu-python-sample 0 --prio $((v)) 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip()); print('prio %-5s kernel_ms %.3f' % ('$v', d['roofline'].get('kernel_ms')))"
  done
done
