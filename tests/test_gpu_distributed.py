"""Multi-GPU path on real devices (BASELINE configs[3] / [4]): one process per GPU under
`torch.distributed.run`, **nccl** (RCCL) backend.  K = 1 always runs (same code path, same
collectives); K = 2 and 8 run when the box has that many GPUs.  Rank 0's gathered records
must equal the single-process result byte for byte, ordered by block index."""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bounded_run  # noqa: E402


def launch_ranks(*args, **kw):
    """A multi-rank launch (torch.distributed.run + a process group): the one kind of child a lost rank
    has hung once without a trace -- repeated once, with a warning (bounded_run.run(retry=True))."""
    return bounded_run.run(*args, retry=True, **kw)


pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _world_sizes():
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 1
    return [k for k in (1, 2, 8) if k <= max(n, 1)]


def _torchrun(k, target, extra, timeout=300):
    # (THRIFTY_SHARDED: what the CLI's own re-launch sets -- marks the ranks as its children)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, THRIFTY_SHARDED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(k),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + target + extra
    res = launch_ranks(cmd, env=env, cwd=ROOT, timeout=timeout, label="torchrun%d" % k)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return res


@pytest.mark.parametrize("k", _world_sizes())
def test_shard_detect_compact_gather_equals_single_process(tmp_path, k):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_detect_worker as w
    from thrifty_amd import _native as F
    n_blocks, seed = 203, 77            # not a multiple of 2 or 8: uneven shards
    out = str(tmp_path / "gathered.npy")
    _torchrun(k, [os.path.join(ROOT, "tests", "dist_detect_worker.py")],
              ["--blocks", str(n_blocks), "--seed", str(seed), "--out", out])
    got = np.load(out).reshape(-1).view(F.RECORD_DTYPE)
    tpl, blocks = w.make_blocks(n_blocks, seed)
    eng = F.Engine(16384, 4096, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=n_blocks)
    rec = eng.detect(blocks, np.arange(n_blocks))[:, 0]
    want = rec[(rec["flags"] & F.FLAG_CORR) != 0]
    assert len(want) > 50
    assert got.tobytes() == want.tobytes()
    assert np.all(np.diff(got["block_idx"]) > 0)


@pytest.mark.parametrize("k", [2, 5, 8])
def test_world_sizes_beyond_the_box_share_gpu0_over_gloo(tmp_path, k):
    """K ranks that all run their shard on GPU 0 and exchange the records over gloo: everything
    of the K-way path except the RCCL transport (uneven shards, empty-ish shards, rank order)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_detect_worker as w
    from thrifty_amd import _native as F
    n_blocks, seed = 37, 78
    out = str(tmp_path / "gathered.npy")
    _torchrun(k, [os.path.join(ROOT, "tests", "dist_detect_worker.py")],
              ["--blocks", str(n_blocks), "--seed", str(seed), "--out", out, "--backend", "gloo"])
    got = np.load(out).reshape(-1).view(F.RECORD_DTYPE)
    tpl, blocks = w.make_blocks(n_blocks, seed)
    rec = F.Engine(16384, 4096, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=n_blocks).detect(
        blocks, np.arange(n_blocks))[:, 0]
    want = rec[(rec["flags"] & F.FLAG_CORR) != 0]
    assert len(want) > 10 and got.tobytes() == want.tobytes()


def _write_case(tmp_path, g, raw=False):
    from thrifty_amd import block_data
    np.save(tmp_path / "template.npy", g["template"])
    (tmp_path / "detector.cfg").write_text(
        "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
        "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
        "template: %s\n" % (tmp_path / "template.npy"))
    text = "# synthetic\n" + "".join(
        block_data.card_line(1000.0 + i, int(g["block_idx"][i]), g["blocks"][i])
        for i in range(len(g["blocks"])))
    (tmp_path / "rx.card").write_text(text)


@pytest.mark.parametrize("k", _world_sizes())
def test_thrifty_detect_gpus_cli_writes_the_single_process_toad(golden, tmp_path, k):
    """`python -m thrifty_amd.detect --gpus K rx.card -o rx.toad` (re-launches itself under
    torch.distributed.run) == the single-process CLI's file, byte for byte; and that file is
    the reference's .toad within the parity tolerances."""
    from thrifty_amd.detect import Detector, detector_cli
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_detector_api import assert_toad_close
    g = golden("c2")
    _write_case(tmp_path, g)
    common = [str(tmp_path / "rx.card"), "--quiet", "-c", str(tmp_path / "detector.cfg")]
    detector_cli(Detector, argv=common + ["-o", str(tmp_path / "one.toad")])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    res = launch_ranks([sys.executable, "-m", "thrifty_amd.detect", "--gpus", str(k)] + common +
                          ["-o", str(tmp_path / "many.toad")], env=env, cwd=ROOT, timeout=300, label="cli%d" % k)
    if k == 1:      # --gpus 1 is the plain single-process CLI; also run it as ONE rank under torchrun
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        _torchrun(1, ["-m", "thrifty_amd.detect"], ["--gpus", "1"] + common + ["-o", str(tmp_path / "rank.toad")])
        assert (tmp_path / "rank.toad").read_text() == (tmp_path / "one.toad").read_text()
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert (tmp_path / "many.toad").read_text() == (tmp_path / "one.toad").read_text()
    assert_toad_close((tmp_path / "one.toad").read_text().strip().split("\n"), g["toad"])


@pytest.mark.parametrize("k,raw", [(8, False), (3, True)])
def test_the_cli_with_more_ranks_than_gpus_rehearsed_over_gloo(golden, tmp_path, k, raw):
    """`python -m thrifty_amd.detect --gpus 8 --dist-backend gloo rx.card -o rx.toad` on the one-GPU
    box: the CLI re-launches itself, eight ranks shard the file (each through the library loop with
    a record sink, one populator and napping waits where CPUs are short), the gather rehearsal and
    the gather run over gloo, rank 0 writes ONE file -- the single-process file, byte for byte."""
    from thrifty_amd.detect import Detector, detector_cli
    g = golden("c2")
    _write_case(tmp_path, g)
    src = str(tmp_path / "rx.card")
    extra = []
    if raw:     # the same blocks' NEW samples as a raw stream (block i = previous tail + these)
        n, h = 16384, 4096
        step = 2 * (n - h)
        with open(tmp_path / "rx.bin", "wb") as f:
            for i in range(60):
                f.write(np.ascontiguousarray(g["blocks"][i % len(g["blocks"])][-step:]).tobytes())
        src, extra = str(tmp_path / "rx.bin"), ["--raw"]
    common = [src, "--quiet", "-c", str(tmp_path / "detector.cfg")] + extra
    detector_cli(Detector, argv=common + ["-o", str(tmp_path / "one.toad")])
    env = dict(os.environ, PYTHONPATH=ROOT)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "THRIFTY_SHARDED"):
        env.pop(key, None)
    res = launch_ranks([sys.executable, "-m", "thrifty_amd.detect", "--gpus", str(k), "--dist-backend", "gloo"]
                          + common + ["-o", str(tmp_path / "many.toad")], env=env, cwd=ROOT, timeout=400,
                          label="cli_gloo%d" % k)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    one, many = (tmp_path / "one.toad").read_text(), (tmp_path / "many.toad").read_text()
    strip = (lambda t: [" ".join(ln.split()[:1] + ln.split()[2:]) for ln in t.strip().split("\n")]) if raw else (lambda t: t)
    assert strip(many) == strip(one) and len(one.strip().split("\n")) >= 15      # (raw: wall-clock stamps differ)


def test_thrifty_detect_raw_gpus_cli_equals_single_process(tmp_path):
    """`--raw --gpus 1` under torchrun (block-range sharding of a raw u8 stream, lead-in blocks on
    rank 0) writes the same .toad as the single-process CLI, and detects the planted bursts."""
    from thrifty_amd import synth
    from thrifty_amd.detect import Detector, detector_cli
    n, h = 16384, 4096
    new = n - h
    tpl = synth.gold_template(10, 2).astype(np.float64)
    np.save(tmp_path / "template.npy", tpl)
    (tmp_path / "detector.cfg").write_text(
        "rxid: 2\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
        "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
        "template: %s\n" % (tmp_path / "template.npy"))
    rng = np.random.default_rng(12)
    nblk = 40
    x = (rng.normal(0, 0.02, nblk * new) + 1j * rng.normal(0, 0.02, nblk * new)).astype(np.complex64)
    ook = 0.3 * (tpl + 1) / 2
    planted = 0
    for b in range(1, nblk - 1, 2):                    # one burst well inside every other block
        pos = b * new + int(rng.integers(2000, new - 2000))
        f = rng.uniform(20, 90)
        ar = np.arange(len(tpl))
        x[pos:pos + len(tpl)] += (ook * np.exp(2j * np.pi * f * (ar + pos) / n)).astype(np.complex64)
        planted += 1
    synth.quantise_iq(x).tofile(tmp_path / "rx.bin")
    common = [str(tmp_path / "rx.bin"), "--raw", "--quiet", "-c", str(tmp_path / "detector.cfg")]
    detector_cli(Detector, argv=common + ["-o", str(tmp_path / "one.toad")])
    _torchrun(1, ["-m", "thrifty_amd.detect"], ["--gpus", "1"] + common + ["-o", str(tmp_path / "rank.toad")])
    def fields(path):       # (a raw stream is stamped with wall-clock time: drop that column)
        return [ln.split()[:1] + ln.split()[2:] for ln in path.read_text().strip().split("\n") if ln]
    one = fields(tmp_path / "one.toad")
    assert fields(tmp_path / "rank.toad") == one
    assert len(one) >= planted - 2 and all(f[0] == "2" for f in one)


# ---------------------------------------------------------------------------------------------
# BASELINE configs[4]: 4 TX templates per block AND the block shard, together
# ---------------------------------------------------------------------------------------------
def _multi_reference(n_blocks, seed, T=4):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_detect_worker as w
    from thrifty_amd import _native as F
    tpls, blocks = w.make_blocks(n_blocks, seed, T)
    rec = F.Engine(16384, 4096, tpls, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=n_blocks).detect(
        blocks, np.arange(n_blocks))
    flat = rec.reshape(-1)
    return flat[(flat["flags"] & F.FLAG_CORR) != 0]


@pytest.mark.parametrize("k,backend", [(k, "nccl") for k in _world_sizes()] + [(2, "gloo"), (5, "gloo"), (8, "gloo")])
def test_multi_template_shard_keeps_block_template_order(tmp_path, k, backend):
    """4 templates x K-way block shard: rank 0's gathered records == the single-process records,
    byte for byte, ordered [block][template] (nccl at the box's device counts; gloo K = 2 / 5 / 8
    with every rank on GPU 0 for the world sizes the box cannot give to RCCL)."""
    from thrifty_amd import _native as F
    n_blocks, seed = 61, 91
    out = str(tmp_path / "gathered.npy")
    _torchrun(k, [os.path.join(ROOT, "tests", "dist_detect_worker.py")],
              ["--blocks", str(n_blocks), "--seed", str(seed), "--out", out, "--templates", "4",
               "--backend", backend])
    got = np.load(out).reshape(-1).view(F.RECORD_DTYPE)
    want = _multi_reference(n_blocks, seed)
    assert len(want) > 25 and len(set(want["template_id"].tolist())) == 4
    assert got.tobytes() == want.tobytes()
    keys = list(zip(got["block_idx"].tolist(), got["template_id"].tolist()))
    assert keys == sorted(keys)


def test_thrifty_detect_templates_gpus_cli_equals_single_process(golden, tmp_path):
    """`python -m thrifty_amd.detect --templates a.npy b.npy c.npy d.npy --gpus 1 rx.card -o out`
    (one rank under torchrun: shard, gather, rank-0 writer) == the single-process CLI's file;
    every line carries rxid then txid; template 0's lines are the single-template run's."""
    from thrifty_amd import synth
    from thrifty_amd.detect import Detector, detector_cli
    g = golden("c2")
    _write_case(tmp_path, g)
    names = []
    for i in range(4):       # template 0 = the golden's own template
        t = g["template"] if i == 0 else synth.gold_template(10, 3 + i).astype(np.float64)
        np.save(tmp_path / ("t%d.npy" % i), t)
        names.append(str(tmp_path / ("t%d.npy" % i)))
    common = [str(tmp_path / "rx.card"), "--quiet", "-c", str(tmp_path / "detector.cfg"), "--templates"] + names
    detector_cli(Detector, argv=common + ["-o", str(tmp_path / "one.toads")])
    _torchrun(1, ["-m", "thrifty_amd.detect"], ["--gpus", "1"] + common + ["-o", str(tmp_path / "rank.toads")])
    one = (tmp_path / "one.toads").read_text()
    assert (tmp_path / "rank.toads").read_text() == one
    lines = [ln.split() for ln in one.strip().split("\n")]
    assert all(ln[0] == "0" and ln[1] in "0123" and len(ln) == 13 for ln in lines)
    detector_cli(Detector, argv=common[:4] + ["-o", str(tmp_path / "single.toad")])
    single = [ln.split() for ln in (tmp_path / "single.toad").read_text().strip().split("\n")]
    tx0 = [ln[:1] + ln[2:] for ln in lines if ln[1] == "0"]
    # (the 4-template kernel is another specialisation: same indices, floats to rounding)
    assert [ln[:3] + ln[4:5] for ln in tx0] == [ln[:3] + ln[4:5] for ln in single]   # rxid, time, block, sample
    assert max(abs(float(a[3]) - float(b[3])) for a, b in zip(tx0, single)) < 1e-5     # soa
    # the non-quiet loop prints the same detections
    detector_cli(Detector, argv=[a for a in common if a != "--quiet"] + ["-o", str(tmp_path / "loud.toads")])
    assert (tmp_path / "loud.toads").read_text() == one
