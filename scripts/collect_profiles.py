#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of scripts/profile_gpu.sh runs from gpurun_out/prof_<tag>/ into
profiles/ (tracked) and rebuild profiles/hbm_traffic.json, which bench.py reads for
`roofline.traffic`:  {config key: {kernel: {fetch_kib_raw, write_kib, bytes_per_launch}}}.

    python scripts/collect_profiles.py r05_c2=c2 r05_t4=c2_t4 r05_c3=c3 r05_sparse=c2_sparse r05_c1=c1 ...
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
if os.path.exists(path):
    try:
        old = json.load(open(path))
        out = {k: v for k, v in old.items() if isinstance(v, dict) and all(isinstance(x, dict) for x in v.values())
               and k in ("c2", "c2_t4", "c3", "c3_t4", "c2_preshift", "c2_sparse", "c2_fullwin", "c1", "c1_sparse")}
    except Exception:
        out = {}
for arg in sys.argv[1:]:
    tag, key = arg.split("=")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    shutil.copy(os.path.join(src, "summary.md"), os.path.join(ROOT, "profiles", tag + "_rocprofv3_summary.md"))
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    out[key] = json.load(open(os.path.join(src, "hbm_traffic.json")))
    sha = out[key].pop("_csrc_sha16", None)
    out[key]["_source"] = {"csrc_sha16": sha, "summary": "profiles/%s_rocprofv3_summary.md" % tag,
                           "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch: FETCH_SIZE doubled for "
                                      "gfx950 as MI355X_MICROARCH.md (HBM section) prescribes"}
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path, "with", sorted(out))
