#!/bin/bash
# dev: VGPR / scratch (spill) use of every kernel in one .hip file of thrifty_amd/csrc
# usage: scripts/resource_usage.sh detect16k.hip [extra hipcc flags]   -- kernels with scratch are marked
F=${1:-detect16k.hip}; shift || true
cd "$(dirname "$0")/../thrifty_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 "$@" -c "$F" -o /tmp/ru_$$.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass-analysis=kernel-resource-usage\]//' |
  paste - - - | awk '{name=$2; sub(/^_ZN3thr[0-9]*/,"",name); printf "%-60s VGPRs %s scratch %s%s\n", substr(name,1,60), $(NF-3), $NF, ($NF!="0" ? "   <-- SPILLS" : "")}'
rm -f /tmp/ru_$$.o
