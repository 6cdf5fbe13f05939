#!/bin/bash
# dev: per-phase timing by early exit (results are garbage for THR_ABLATE != 0)
for a in 0 1 2 3 4 5 6; do
  THR_ABLATE=$a python bench.py --steps 16 --warmup 2 --batch 8192 --cpu-seconds 0 --profile-kernels 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$a', {k: round(v,4) for k,v in d['roofline']['all_kernels_ms'].items()})"
done
