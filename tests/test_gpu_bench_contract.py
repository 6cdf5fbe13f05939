"""bench.py's output contract, on a shortened run: ONE JSON line on stdout carrying the driver's
keys, a recomputable `roofline`, `cpu_baseline`, and the legs of the other single-GPU configs."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bounded_run  # noqa: E402


def launch_ranks(*args, **kw):
    """A multi-rank launch (torch.distributed.run + a process group): the one kind of child a lost rank
    has hung once without a trace -- repeated once, with a warning (bounded_run.run(retry=True))."""
    return bounded_run.run(*args, retry=True, **kw)


pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LEG_KEYS = ("value", "kernel", "frac", "avg_launch_ms", "blocks_per_launch", "algorithmic_bytes_per_block",
            "traffic_over_algorithmic", "pipeline_traffic_over_algorithmic", "pipeline_frac", "valu_frac",
            "traffic_stale")


def test_default_run_line_has_every_contract_key():
    detail_path = os.path.join(ROOT, "gpurun_out", "bench_detail_c2_n1.json")
    if os.path.exists(detail_path):
        os.remove(detail_path)
    res = bounded_run.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                           "--min-seconds", "0.2", "--leg-seconds", "0.1", "--cpu-seconds", "0.5",
                           "--cpu-procs", "2", "--card-blocks", "256"], cwd=ROOT, timeout=600, label="bench_default")
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    # the driver keeps the last 8 KB of stdout: the whole line must fit, the flat summary at its end
    assert len(lines[0]) < 7000, len(lines[0])
    d = json.loads(lines[0])
    assert list(d)[-1] == "summary"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 3 and d["n_gpus"] == 1 and d["unit"] == "blocks/s" and d["vs_baseline"] is None
    assert d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = blocks of exactly K steps / their time
    blocks = d["steps"] * d["config"]["blocks_per_step_per_gpu"]
    assert blocks == d["blocks_timed"]
    assert abs(d["value"] - blocks / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    assert d["timed_region_s"] >= 0.15
    assert d["compact_ms"] > 0 and d["gather_ms"] >= 0
    assert len(d["config"]["per_rank_seconds"]) == 1 and 0 < d["config"]["per_rank_seconds"][0] <= d["timed_region_s"]
    # the main roofline object: recomputable, flat scalars only (the driver's record drops nested ones)
    r = d["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_correlate_4k"
    want = r["algorithmic_bytes_per_block"] * r["blocks_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(r["achieved"] - want) <= 1e-6 * want and abs(r["frac"] - want / 8000.0) <= 1e-9
    assert abs(r["pipeline_frac"] - d["value"] * r["algorithmic_bytes_per_block"] / 8e12) <= 1e-9
    assert r["pipeline_frac"] < r["frac"] < 1 and 0 < r["valu_frac"] < 1
    for key in ("traffic", "traffic_over_algorithmic", "pipeline_traffic_over_algorithmic", "traffic_stale",
                "traffic_source"):
        assert key in r, key
    # every leg: the agreed compact keys, recomputable fraction, and the same value in the summary
    assert set(d["configs"]) == {"c3", "t4", "sparse", "fullwin", "c3t4", "c1", "c1_sparse"}
    for name, leg in d["configs"].items():
        for key in LEG_KEYS:
            assert key in leg, (name, key)
        assert all(not isinstance(v, (dict, list)) for v in leg.values()), name
        want = leg["algorithmic_bytes_per_block"] * leg["blocks_per_launch"] / (leg["avg_launch_ms"] * 1e-3) / 8e12
        assert abs(leg["frac"] - want) <= 1e-9 and leg["value"] > 0, name
        assert d["summary"][name] == round(leg["value"])
    assert d["summary"]["c2"] == round(d["value"])
    assert d["configs"]["c3"]["kernel"] == "k_correlate_seg" and d["configs"]["c3"]["handles_per_gpu"] == 1
    assert d["configs"]["c3t4"]["templates"] == 4
    assert d["configs"]["c1"]["kernel"] == "k_correlate" and d["configs"]["c1_sparse"]["kernel"] == "k_carrier"
    assert d["configs"]["c3"]["parity_mismatches"] == 0 and d["configs"]["c3"]["parity_checked"] == 2048
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["parity_mismatches"] == 0 and cb["value"] > 0
    assert all(not isinstance(v, (dict, list)) for v in cb.values())
    assert cb["all_cores_value"] > 0 and cb["card_to_toad_outputs_agree"] is True
    assert d["summary"]["card_to_toad"] == round(cb["card_to_toad_gpu_blocks_per_s"])
    assert d["summary"]["raw_to_toad"] == round(cb["card_to_toad_raw_gpu_blocks_per_s"]) > 0
    # what left the line is in the detail file (and on stderr)
    assert "bench detail: {" in res.stderr
    det = json.load(open(detail_path))
    assert det["configs"]["fullwin"]["roofline_detail"]["all_kernels_ms"]["k_carrier"] > 0
    assert "every bin" in det["configs"]["fullwin"]["workload"]
    assert det["main"]["roofline_detail"]["compute"]["peak_tflops"] > 100
    assert det["card_to_toad"]["gpu_loop_stats"]["batches"] >= 1


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
    # attribution of a sub-linear curve: every rank's own clock, its compaction and its gather
    for key in ("per_rank_seconds", "per_rank_compact_ms", "per_rank_gather_ms"):
        assert len(c[key]) == world and all(v >= 0 for v in c[key]), key
    assert max(c["per_rank_seconds"]) <= d["timed_region_s"] + 1e-6
    assert c["gather_method"] in ("gather", "all_gather")
    assert c["rank_env"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    assert d["gather_ms"] > 0 and d["compact_ms"] > 0


SHORT = ["--steps", "2", "--warmup", "1", "--min-seconds", "0.2", "--legs", "none", "--cpu-seconds", "0",
         "--resident-blocks", "65536"]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gpus_n_launches_itself_and_the_whole_n_rank_body_runs_over_gloo(world):
    """`python bench.py --gpus N` with no torchrun around it: the script starts its own ranks.
    --dist-backend gloo puts every rank on cuda:0, so pre-flight, the step-size broadcast, the MAX
    all-reduce of the time and the record gather all run on this one-GPU box."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = launch_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world),
                           "--dist-backend", "gloo"] + SHORT, cwd=ROOT, timeout=500, env=env, label="bench_gloo%d" % world)
    d = _one_line(res)
    _check_dist_line(d, world, "gloo")
    assert res.stdout.strip().split("\n")[-1].startswith("{")       # the contract line is the LAST line
    assert "pre-flight ok: backend gloo, %d rank(s)" % world in res.stderr
    # the environment of every rank and the gather rehearsal (uneven counts, an empty rank) are on record
    for r in range(world):
        assert "pre-flight env: rank %d " % r in res.stderr
    assert res.stderr.count("HSA_ENABLE_IPC_MODE_LEGACY=0") >= world
    assert "pre-flight gather selftest ok: method gather" in res.stderr


def test_strong_scaling_splits_one_gpus_job_over_the_ranks():
    """--scaling strong: the blocks one GPU would run, in total (BASELINE configs[3] as written)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    lines = {}
    for world in (1, 2):
        res = launch_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world),
                               "--dist-backend", "gloo", "--scaling", "strong"] + SHORT + ["--steps", "4"],
                              cwd=ROOT, timeout=500, env=env, label="bench_strong%d" % world)
        lines[world] = _one_line(res)
        assert lines[world]["scaling"] == "strong"
    one, two = lines[1], lines[2]
    # per-rank residency halves, and so does the per-rank step (to within the rounding of R)
    assert "32768 blocks/GPU" in two["config"]["workload"] and "65536 blocks/GPU" in one["config"]["workload"]
    # (the one-GPU step is calibrated per run -- and the rehearsal's two ranks share GPU 0 --, so each
    # line is checked against its own calibration)
    for world, d in lines.items():
        c = d["config"]
        assert c["launch_batches_per_step"] == -(-c["launch_batches_per_step_one_gpu"] // world)
    assert two["blocks_timed"] == 2 * two["steps"] * two["config"]["blocks_per_step_per_gpu"]


def test_one_rank_under_torchrun_runs_the_same_body_over_rccl():
    """The driver's launch line at N = 1: torch.distributed.run, nccl (= RCCL) process group,
    collectives on device tensors."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = launch_ranks([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(port),
                           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SHORT, cwd=ROOT, timeout=500,
                          label="bench_rccl1")
    d = _one_line(res)
    _check_dist_line(d, 1, "nccl")
    assert "pre-flight env: rank 0 " in res.stderr and "HSA_ENABLE_IPC_MODE_LEGACY=0" in res.stderr
    assert "pre-flight gather selftest ok" in res.stderr


def test_more_ranks_than_gpus_is_refused_legibly():
    import torch
    world = torch.cuda.device_count() + 1
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = launch_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + SHORT,
                          cwd=ROOT, timeout=400, env=env, label="bench_refused")
    assert res.returncode != 0
    assert "one GPU per rank" in res.stderr
