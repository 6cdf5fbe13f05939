"""bench.py's output contract, on a shortened run: ONE JSON line on stdout carrying the driver's
keys, a recomputable `roofline`, `cpu_baseline`, and the legs of the other single-GPU configs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_run_line_has_every_contract_key():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--min-seconds", "0.2", "--leg-seconds", "0.1", "--cpu-seconds", "0.5",
                          "--cpu-procs", "2", "--card-blocks", "256"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 3 and d["n_gpus"] == 1 and d["unit"] == "blocks/s" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = blocks of exactly K steps / their time
    blocks = d["steps"] * d["config"]["blocks_per_step_per_gpu"]
    assert blocks == d["blocks_timed"]
    assert abs(d["value"] - blocks / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    assert d["timed_region_s"] >= 0.15
    for name, obj in [("main", d)] + list(d["configs"].items()):
        r = obj["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0, name
        want = r["algorithmic_bytes_per_block"] * r["blocks_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
        assert abs(r["achieved"] - want) <= 1e-6 * want and abs(r["frac"] - want / 8000.0) <= 1e-9, name
        assert obj["value"] > 0
    assert set(d["configs"]) == {"c3", "t4", "sparse"}
    assert d["configs"]["c3"]["cpu_baseline"]["parity_mismatches"] == 0
    assert d["configs"]["c3"]["cpu_baseline"]["parity_checked"] == 2048
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["parity_mismatches"] == 0 and cb["value"] > 0
    assert cb["card_to_toad"]["first_lines_agree_on_rxid_time_block_sample_bin"]
