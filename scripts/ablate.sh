#!/bin/bash
# dev: per-phase timing by early exit (results are garbage for THR_ABLATE != 0)
# needs a library built with THR_EXTRA_CFLAGS=-DTHR_DEV_ABLATE python -m thrifty_amd.build --force
# 1..3: k_carrier (full kernel; use THR_NO_PRUNE=1) after pass 1/2/3; 11..16: k_correlate after P1,P2,P3,A,B,C
for a in ${@:-0 11 12 13 14 15 16}; do
  THR_ABLATE=$a python bench.py --steps 16 --warmup 2 --batch 8192 --cpu-seconds 0 --profile-kernels 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$a', {k: round(v,4) for k,v in d['roofline']['all_kernels_ms'].items()})"
done
