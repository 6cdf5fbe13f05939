run() { python bench.py --config c3 --cpu-seconds 0 --steps 24 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), {k: round(v,4) for k,v in d['roofline']['all_kernels_ms'].items()})"; }
export THRIFTY_HIP_LIB=$PWD/ab/long2.so
THR_LONG_OVERLAP=0 run "ovl0 default"
THR_LONG_OVERLAP=0 THR_LONG_BATCH=4096 run "ovl0 batch4096"
run "ovl1 default(chunk128)"
THR_LONG_BATCH=4096 run "ovl1 batch4096"
THR_LONG_BATCH=4096 THR_LONG_CHUNK=256 run "ovl1 batch4096 chunk256"
THR_LONG_BATCH=4096 THR_LONG_CHUNK=64 run "ovl1 batch4096 chunk64"
THR_LONG_BATCH=4096 THR_LONG_CHUNK=512 run "ovl1 batch4096 chunk512"
