#!/bin/bash
# dev: variants of detect16k_sec.hip alone (extra -D flags), each linked against the default build's other
# objects -> ab/<name>.so.    scripts/sec_ab.sh name1 "-DFOO=1" name2 "-DBAR" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
C=thrifty_amd/csrc
OBJS=$(python -c "from thrifty_amd import build as b; print(' '.join('$C/' + s.replace('.hip', '.o') for s in b.SOURCES if s != 'detect16k_sec.hip'))")
while [ $# -ge 2 ]; do
  N=$1; F=$2; shift 2
  (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None $F -c $C/detect16k_sec.hip -o ab/$N.o &&
   hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$N.so $OBJS ab/$N.o && echo "ab/$N.so [$F]") &
done
wait
