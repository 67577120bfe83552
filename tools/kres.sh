#!/bin/bash
# Registers and spills of every kernel of one translation unit: tools/kres.sh FILE.hip [extra hipcc flags]
cd "$(dirname "$0")/../tagdigger_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include "$@" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  sed 's/ *\[-Rpass[^]]*\]//' |
  awk '/Function Name:/ {name=$NF} / VGPRs: / {v=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF} /ScratchSize/ {sc=$NF} /LDS Size/ {print name, "VGPRs", v, "VGPR-spill", vs, "SGPR-spill", ss, "scratch", sc}' | c++filt
