#!/usr/bin/env python3
"""Dev aid: instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing, per basic
block (blocks with >= MIN instructions) -- the VALU count is what k_correlate's time follows.
    python scripts/asm_mix.py file.s <symbol substring> [min block size] [--ops]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 100
ops = "--ops" in sys.argv
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and sym in l.split(":")[0]][0]
end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith("s_endpgm")][0]
blocks, cur = [], ("entry", [])
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
        continue
    s = l.strip()
    if s and not s.startswith(";") and not s.startswith("."):
        cur[1].append(s.split()[0])
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter()
    for i in ins:
        k = ("pk" if i.startswith("v_pk_") else "valu" if i.startswith("v_") else "nop" if i == "s_nop" else
             "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem")
        c[k] += 1
    tot.update(c)
    if len(ins) >= minsz:
        print(name, len(ins), dict(c))
        if ops:
            for k, v in collections.Counter(ins).most_common(45):
                print("     %-28s %d" % (k, v))
print("total", dict(tot))
for l in lines[end:end + 400]:
    if "NumVgprs" in l or "ScratchSize" in l:
        print(l.strip())
