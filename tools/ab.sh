#!/bin/bash
# A/B of two builds of libtagdig on one box, alternating: tools/ab.sh <libA.so> <libB.so> [bench args...]
# prints the headline pass's kernel_ms per run (HIP events around the count kernels).
A=$1; B=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS="--steps 10 --warmup 2 --cpu-sample 0 --tier-reads 0 --other-configs= --traffic off --oracle-sample 0 $*"
for round in 1 2 3; do
  for lib in $A $B; do
    TAGDIG_LIB=$ROOT/tagdigger_amd/$lib python3 $ROOT/bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.readline())
print('$lib', 'kernel_ms %.3f min %.3f  prog %.3f  exact %s' % (o['roofline']['kernel_ms'], o['roofline']['kernel_ms_min'], (o.get('progress_windows') or {}).get('kernel_ms') or 0, o['check']['bit_exact_vs_expected']))"
  done
done
