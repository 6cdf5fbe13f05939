"""The documents quote measurements; the measurements live in `profiles/` and in the driver's
`BENCH_rNN.json`.  Round 4's README and DESIGN claimed a `.card` -> `.toad` rate (0.85-1.0 M blocks/s,
from a side script) that no saved default run of `bench.py` showed (0.70-0.74 M).  The tables of
README.md ("Measured") and DESIGN.md (section 6) are therefore held to the record of the saved default
run, `profiles/r06_bench_default_run.json`: every `summary` key has a row, every row's blocks/s is the
record's (to the rounding of the table), and the per-kernel figures of DESIGN's table are the
record's.  README's last column quotes the DRIVER's run of the previous round's tree and is held to
`BENCH_r05.json` -- the file the README names, not "the latest": the driver writes this round's after
the tree is final."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "profiles", "r06_bench_default_run.json")
DRIVER = os.path.join(ROOT, "BENCH_r05.json")


def table_rows(path, n_cols):
    """{key: [cell, ...]} of the markdown table rows whose first cell is a back-ticked summary key."""
    rows = {}
    for line in open(os.path.join(ROOT, path), encoding="utf-8"):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        m = re.fullmatch(r"`(\w+)`", cells[0]) if cells else None
        if m and len(cells) == n_cols and re.search(r"\d M\b", line):
            rows[m.group(1)] = cells
    return rows


def mega(cell):
    m = re.search(r"(\d+(?:\.\d+)?) M\b", cell)
    assert m, cell
    return float(m.group(1)) * 1e6


def test_readme_and_design_tables_are_the_saved_default_run():
    rec = json.load(open(RECORD))
    summary = rec["summary"]
    assert rec["steps"] == 20 and rec["warmup"] == 5 and rec["n_gpus"] == 1      # the driver's command line
    for path, n_cols, col in (("README.md", 4, 2), ("DESIGN.md", 4, 1)):
        rows = table_rows(path, n_cols)
        assert set(rows) == set(summary), (path, sorted(set(rows) ^ set(summary)))
        for key, cells in rows.items():
            got, want = mega(cells[col]), summary[key]
            assert abs(got - want) <= 0.006 * want + 5e3, (path, key, got, want)   # two decimals of a rounded M


def driver_summary(path):
    """The flat `summary` object of the JSON line in the driver's record of its bench run."""
    rec = json.load(open(path))
    m = re.search(r'"summary": (\{[^}]*\})', rec["run"]["stdout_tail"])
    assert m, "no summary in %s" % path
    return json.loads(m.group(1))


def test_readme_driver_column_is_the_drivers_record():
    if not os.path.exists(DRIVER):       # (a tree without the driver's records)
        return
    assert os.path.basename(DRIVER) in open(os.path.join(ROOT, "README.md"), encoding="utf-8").read()
    theirs = driver_summary(DRIVER)
    rows = table_rows("README.md", 4)
    for key, cells in rows.items():
        if key in theirs:
            got = mega(cells[3])
            assert abs(got - theirs[key]) <= 0.006 * theirs[key] + 5e3, (key, got, theirs[key])
        else:
            assert cells[3].startswith("—"), (key, cells[3])


def test_design_kernel_figures_are_the_records():
    rec = json.load(open(RECORD))
    legs = dict(rec["configs"], c2=dict(rec["roofline"], value=rec["value"]))
    rows = table_rows("DESIGN.md", 4)
    for key, leg in legs.items():
        cell, pipe = rows[key][2], rows[key][3]
        assert "`%s`" % {"k_carrier": "k_carrier_pruned"}.get(leg["kernel"], leg["kernel"]) in cell, (key, cell)
        ms, frac, ratio = (float(v) for v in re.search(r"(\d+\.\d+) ms, (\d\.\d+), (\d\.\d+)", cell).groups())
        assert abs(ms - leg["avg_launch_ms"]) <= 0.002 and abs(frac - leg["frac"]) <= 0.001, (key, cell)
        assert abs(ratio - leg["traffic_over_algorithmic"]) <= 0.002, (key, cell)
        assert abs(float(pipe) - leg["pipeline_frac"]) <= 0.001, (key, pipe)
        assert leg["traffic_stale"] is False


def test_no_document_still_quotes_the_superseded_file_rates():
    """The claims the round-4 review found no record for, and the numbers they replaced."""
    stale = ["0.85–1.0 M", "0.85–0.9 M blocks/s** from a 65536-line", "486 k blocks/s, raw file",
             "the leg itself runs 16384 lines"]
    for path in ("README.md", "DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, path), encoding="utf-8").read()
        for phrase in stale:
            assert phrase not in text, (path, phrase)
    # (HISTORY.md keeps them, as history)


def test_design_is_the_current_design_only():
    """The round-5 review: DESIGN.md had grown to 1138 lines of round-by-round narrative.  It holds the
    current design (<= 400 lines); the narrative is HISTORY.md, the experiments profiles/README.md."""
    lines = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().split("\n")
    assert len(lines) <= 400, len(lines)
    assert os.path.exists(os.path.join(ROOT, "HISTORY.md"))
