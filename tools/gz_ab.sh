#!/bin/bash
# The device gzip decoder's stage times under variant libraries and options, one after the other in one gpurun call:
#   tools/gz_ab.sh TAG "lib[:TD_OPTS]" ...      e.g.  tools/gz_ab.sh ab libtagdig.so libtagdig_bperm.so "libtagdig.so:gz_gpu_terr_kb=64"
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for arm in "$@"; do
  lib=${arm%%:*}; opts=""; [ "$arm" != "$lib" ] && opts=${arm#*:}
  echo "== $lib  TD_OPTS='$opts'" | tee -a "$OUT/ab.txt"
  TAGDIG_LIB=$ROOT/tagdigger_amd/$lib TD_OPTS="$opts" TAGDIG_INFLATE_STATS=1 timeout -k 10 150 python3 "$ROOT/tools/gz_tier.py" > "$OUT/arm.out" 2> "$OUT/arm.err" || { tail -5 "$OUT/arm.err"; exit 1; }
  grep -a "^call" "$OUT/arm.out" | tee -a "$OUT/ab.txt"
  grep -a "gz_gpu_inflate: segment" "$OUT/arm.err" | tail -1 | tee -a "$OUT/ab.txt"
done
