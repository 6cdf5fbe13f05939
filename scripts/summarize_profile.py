#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs produced by scripts/profile_gpu.sh into markdown."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    # the long-block correlate kernel exists in a fused form (sub-transforms + combination, does
    # the batch) and a two-kernel form (launched too, returns at once for large batches): 4th
    # template argument
    m = re.search(r"k_correlate_sub<\w+, \w+, \w+, (\w+)", name)
    if m:
        return "k_correlate_sub" if m.group(1) in ("true", "1") else "k_correlate_sub(two-kernel form)"
    # k_correlate<FMT, STD, MULTI, DUMP, RLO, RHI, SEG>: SEG = the (block, section) items of long blocks
    m = re.search(r"k_correlate<[^>]*, (\w+)>", name)
    if m and m.group(1) in ("true", "1"):
        return "k_correlate_seg"
    if "k_correlate_4k" in name:     # block_len 16384, short template: the (block, 4096-sample section) items
        return "k_correlate_4k"
    for k in ("k_carrier_pruned", "k_carrier_dit", "k_carrier_sub_pruned", "k_carrier_sub", "k_carrier_small",
              "k_carrier", "k_select_dit", "k_select", "k_fit_preshift", "k_fit", "k_finish",
              "k_correlate_sub", "k_correlate_small", "k_correlate", "k_combine", "k_preshift",
              "k_compact_count", "k_compact_scan", "k_compact_scatter", "k_b64_decode"):
        if k in name:
            return k
    return name[:60]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("# rocprofv3 summary (%s)\n" % os.path.basename(root.rstrip("/")))
for f in find("*kernel_stats.csv"):
    print("## kernel stats (`--kernel-trace --stats`)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print("| %s | %s | %.3f | %.1f | %.1f | %.1f | %s |" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
            float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            r["Percentage"]))
    print()

# counters: average per dispatch per kernel
acc = defaultdict(lambda: defaultdict(list))
for f in find("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    print("## PMC counters (mean per dispatch)\n")
    names = sorted({c for k in acc for c in acc[k]})
    print("| counter | " + " | ".join(sorted(acc)) + " |")
    print("|---|" + "---|" * len(acc))
    for c in names:
        row = []
        for k in sorted(acc):
            v = acc[k].get(c)
            row.append("%.4g" % (sum(v) / len(v)) if v else "-")
        print("| %s | %s |" % (c, " | ".join(row)))
    print()
    print("FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced")
    print("reads by 2x (MI355X_MICROARCH.md, HBM section) -- see DESIGN.md for the corrected figure.")
    import json
    # effective clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / duration of the same
    # dispatches (timestamps of the GRBM pass's own kernel trace)
    dur = defaultdict(list)
    for f in find("*kernel_trace.csv"):
        if "pmc_grbm" not in f:
            continue
        for r in csv.DictReader(open(f)):
            try:
                dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            except (KeyError, ValueError):
                pass
    clocks = {}
    for k in acc:
        g = acc[k].get("GRBM_GUI_ACTIVE")
        if g and dur.get(k):
            clocks[k] = (sum(g) / len(g)) / 8.0 / (sum(dur[k]) / len(dur[k]))     # cycles per ns = GHz
    if clocks:
        print()
        print("## effective clock under PMC collection (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration)\n")
        print("| kernel | GHz | mean duration us (GRBM pass) |")
        print("|---|---|---|")
        for k in sorted(clocks):
            print("| %s | %.3f | %.1f |" % (k, clocks[k], sum(dur[k]) / len(dur[k]) / 1e3))
    traffic = {}
    for k in acc:
        if k.startswith("k_") and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
            w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
            traffic[k] = {"fetch_kib_raw": f, "write_kib": w, "bytes_per_launch": int((2 * f + w) * 1024)}
            if k in clocks:
                traffic[k]["effective_clock_ghz"] = clocks[k]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from thrifty_amd import build
    traffic["_csrc_sha16"] = build.csrc_hash()      # the kernel sources these counters belong to
    json.dump(traffic, open(os.path.join(root, "hbm_traffic.json"), "w"), indent=1)
