#!/bin/bash
# dev: counter passes of one command, per-kernel means -> stdout.   scripts/pmc_quick.sh "<cmd>" "<counters pass 1>" ["<pass 2>" ...]
CMD=$1; shift
export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1)); O=/tmp/pmcq_$i; rm -rf $O
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O -o p -- $CMD > /tmp/pmcq_$i.log 2>&1
  python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file in", sys.argv[1]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "correlate" not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", len(next(iter(d.values()))))
PY
done
