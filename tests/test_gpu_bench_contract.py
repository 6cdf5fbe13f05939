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
    assert set(d["configs"]) == {"c3", "t4", "sparse", "fullwin", "c3t4"}
    assert d["configs"]["fullwin"]["roofline"]["all_kernels_ms"]["k_carrier"] > 0
    assert "every bin" in d["configs"]["fullwin"]["workload"]
    assert d["configs"]["c3"]["roofline"]["kernel"] == "k_correlate_seg"
    assert d["configs"]["c3t4"]["templates"] == 4
    # traffic figures come from the committed PMC passes and say which sources they belong to
    for obj in [d] + list(d["configs"].values()):
        r = obj["roofline"]
        for key in ("traffic", "pipeline_traffic", "pipeline_traffic_over_algorithmic", "traffic_stale",
                    "traffic_source"):
            assert key in r, key
        if r["pipeline_traffic"] is not None:
            assert r["pipeline_traffic"] >= r["traffic"] > 0
            assert isinstance(r["traffic_stale"], bool)
    assert d["configs"]["c3"]["cpu_baseline"]["parity_mismatches"] == 0
    assert d["configs"]["c3"]["cpu_baseline"]["parity_checked"] == 2048
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["parity_mismatches"] == 0 and cb["value"] > 0
    assert cb["card_to_toad"]["first_lines_agree_on_rxid_time_block_sample_bin"]


def _one_line(res):
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def _check_dist_line(d, world, backend):
    c = d["config"]
    assert d["n_gpus"] == world and c["parallelism"] == "block-shard x%d" % world
    assert c["dist_backend"].startswith(backend)
    assert d["blocks_timed"] == world * d["steps"] * c["blocks_per_step_per_gpu"]
    # whole-job value = blocks of ALL ranks / the slowest rank's time
    assert abs(d["value"] - d["blocks_timed"] / d["timed_region_s"]) <= 1e-6 * d["value"]
    assert abs(d["ms_per_step"] * d["steps"] * 1e-3 - d["timed_region_s"]) <= 1e-9 + 1e-6 * d["timed_region_s"]
    # the gather brought every rank's detections to rank 0 (dense mix: nearly every block detects)
    assert len(c["detections_per_rank"]) == world
    assert sum(c["detections_per_rank"]) == c["detections_gathered"]
    assert min(c["detections_per_rank"]) > 0.9 * 65536


SHORT = ["--steps", "2", "--warmup", "1", "--min-seconds", "0.2", "--legs", "none", "--cpu-seconds", "0",
         "--resident-blocks", "65536"]


@pytest.mark.parametrize("world", [2, 3])
def test_gpus_n_launches_itself_and_the_whole_n_rank_body_runs_over_gloo(world):
    """`python bench.py --gpus N` with no torchrun around it: the script starts its own ranks.
    --dist-backend gloo puts every rank on cuda:0, so pre-flight, the step-size broadcast, the MAX
    all-reduce of the time and the record gather all run on this one-GPU box."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world),
                          "--dist-backend", "gloo"] + SHORT,
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    d = _one_line(res)
    _check_dist_line(d, world, "gloo")
    assert "pre-flight ok: backend gloo, %d rank(s)" % world in res.stderr


def test_one_rank_under_torchrun_runs_the_same_body_over_rccl():
    """The driver's launch line at N = 1: torch.distributed.run, nccl (= RCCL) process group,
    collectives on device tensors."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SHORT,
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    d = _one_line(res)
    _check_dist_line(d, 1, "nccl")


def test_more_ranks_than_gpus_is_refused_legibly():
    import torch
    world = torch.cuda.device_count() + 1
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + SHORT,
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode != 0
    assert "one GPU per rank" in res.stderr
