#!/bin/bash
# timing-only experiments on the main pass (counts are wrong for non-zero masks)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
one() {
  python3 "$ROOT/bench.py" --steps 5 --warmup 1 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 --no-check "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', 'ms %.3f min %.3f' % (r['kernel_ms'], r['kernel_ms_min']))"
}
for m in 0 4 2 1 64 128 256 320 384; do one --debug-ablate $m; done
one --blocks-per-cu 3
one --blocks-per-cu 2
one --nt 0
one --table-load 25
one --table-load 90
