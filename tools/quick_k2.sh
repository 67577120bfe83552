#!/bin/bash
# quick correctness + timing round for the main-pass kernels (one gpurun call): parity subset, then the bench per option set
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
TD_FUZZ_SECONDS=${FUZZ:-25} timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/quick_tests.log 2>&1
tail -4 gpurun_out/quick_tests.log
one() {
  python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --tier-reads 0 --oracle-sample 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', 'ms %.3f min %.3f frac %.3f' % (r['kernel_ms'], r['kernel_ms_min'], r['frac']), d['check'] and d['check']['bit_exact_vs_expected'])"
}
one --opt kernel=2
one --opt kernel=2 --opt tile_kb2=32
one --opt kernel=1
one --opt kernel=2 --opt hot_cache=0
one --opt kernel=2 --skew 1.0 --steps 5
one --opt kernel=2 --skew 1.5 --steps 5
