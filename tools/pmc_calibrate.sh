#!/bin/bash
# Calibrate FETCH_SIZE on this kernel's own access pattern: the same launch with phase D disabled
# (stream only: a known 43.8 GB) and with the count atomics disabled (stream + table probes).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_cal
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for ab in 4 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/ab${ab}_$c" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-sample 0 --no-check --debug-ablate $ab > "$OUT/ab${ab}_$c.log" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for ab in (4, 1, 0):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s/ab%d_%s/**/*counter_collection.csv" % (out, ab, c), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_fast" in r["Kernel_Name"] and r["Counter_Name"] == c]
        big = [v for v in vals if v > 1e5]
        print("ablate=%d %s: main-dispatch mean = %.3f GB (n=%d)" % (ab, c, sum(big) / max(1, len(big)) * 1024 / 1e9, len(big)))
PY
