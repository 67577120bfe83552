#!/bin/bash
# timing of option sets (checked against the expected matrix): one line each
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
one() {
  python3 "$ROOT/bench.py" --steps 8 --warmup 2 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', 'ms %.3f min %.3f frac %.3f' % (r['kernel_ms'], r['kernel_ms_min'], r['frac']), d['check'] and d['check']['bit_exact_vs_expected'], 'fixups', r['fixup_queue'])"
}
while read -r line; do [ -n "$line" ] && one $line; done
