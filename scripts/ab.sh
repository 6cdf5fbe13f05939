#!/bin/bash
# dev: interleaved A/B of two builds on the same box: scripts/ab.sh ab/old.so ab/new.so [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do for L in $A $B; do
  THRIFTY_HIP_LIB=$PWD/$L python bench.py --steps 64 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), {k: round(v,4) for k,v in d['roofline']['all_kernels_ms'].items()})"
done; done
