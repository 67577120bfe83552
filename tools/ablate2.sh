#!/bin/bash
# timing-only ablations of the main pass (results are wrong with a non-zero mask): kernel ms per mask
#   usage: tools/ablate2.sh [bench args...]     masks: 4 no phase D, 2 no tag probe, 1 no count update
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for m in 0 4 2 1; do
  python3 "$ROOT/bench.py" --steps 5 --warmup 1 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 --no-check --debug-ablate $m "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('ablate $m $*', 'ms %.3f min %.3f' % (r['kernel_ms'], r['kernel_ms_min']))"
done
