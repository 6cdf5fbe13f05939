#!/bin/bash
# dev: per-kernel times of several library builds on one box, interleaved:
#   scripts/abx.sh "<bench args>" ab/a.so ab/b.so ...   (env passes through, e.g. THR_NO_PRUNE=1 with a -DTHR_DEV build)
ARGS=$1; shift
for r in 1 2; do for L in "$@"; do
  THRIFTY_HIP_LIB=$PWD/$L python bench.py --steps 24 --warmup 2 --cpu-seconds 0 --streams 1 --profile-kernels 1 --legs none --min-seconds 0.5 $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s' % '$L', round(d['value']), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'], 4))"
done; done
