"""world_size-2 CPU test (gloo) of the only collective on the path: the gather of
detection records to rank 0, plus the block-range sharding."""
import os
import socket

import numpy as np
import pytest

from thrifty_amd import parallel


def test_shard_range_covers_everything():
    for n, w in [(10, 2), (10, 3), (7, 8), (1 << 20, 8), (0, 4)]:
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, counts, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = counts[rank]
        local = torch.zeros((n, 64), dtype=torch.uint8)
        # block_idx field (first 8 bytes) = global position, as a shard would produce
        lo = sum(counts[:rank])
        idx = torch.arange(lo, lo + n, dtype=torch.int64)
        local[:, :8] = idx.view(torch.uint8).view(n, 8) if n else local[:, :8]
        local[:, 8] = rank + 1
        out = parallel.gather_records(local, world, rank)
        if rank == 0:
            ret.put(out.numpy().copy())
        else:
            assert out.shape == (0, 64)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(5, 3), (0, 4), (7, 0)])
def test_gather_records_two_ranks(counts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (sum(counts), 64)
    idx = got[:, :8].copy().view(np.int64).reshape(-1)
    assert np.array_equal(idx, np.arange(sum(counts)))          # globally ordered
    assert list(got[:, 8]) == [1] * counts[0] + [2] * counts[1]  # rank order


def _selftest_worker(rank, world, port, break_gather, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if break_gather == "raises":
            def broken(*a, **k):
                raise RuntimeError("backend has no gather")
            dist.gather = broken                 # (a backend without the operation: it raises on every rank)
        if break_gather == "garbles" and rank == 0:
            real = dist.gather

            def garbled(tensor, gather_list=None, dst=0, group=None):
                real(tensor, gather_list, dst=dst, group=group)
                gather_list[1].zero_()           # rank 0 receives something else than was sent
            dist.gather = garbled
        if break_gather == "device":
            # neither collective of the run's backend moves the record buffer; a second group does
            real_g, real_ag = dist.gather, dist.all_gather

            def no_gather(tensor, gather_list=None, dst=0, group=None):
                if group is None:
                    raise RuntimeError("no gather on the device backend")
                return real_g(tensor, gather_list, dst=dst, group=group)

            def no_records(tensor_list, tensor, group=None):
                if group is None and tensor.dim() == 2:
                    raise RuntimeError("all_gather refuses the record buffer")
                return real_ag(tensor_list, tensor, group=group)
            dist.gather, dist.all_gather = no_gather, no_records
        method = parallel.gather_selftest(world, rank, torch.device("cpu"))
        # ... and the run's own gather then takes the method the rehearsal settled on
        counts = [4, 0, 9][:world]
        local = torch.full((counts[rank], 64), rank + 1, dtype=torch.uint8)
        out = parallel.gather_records(local, world, rank)
        ret.put((rank, method, out.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,break_gather,want", [(2, None, "gather"), (3, None, "gather"),
                                                     (3, "raises", "all_gather"), (3, "garbles", "all_gather"),
                                                     (3, "device", "host")])
def test_gather_selftest_settles_the_method_collectively(world, break_gather, want):
    """The pre-flight rehearsal of the record gather (uneven counts, one empty rank): a backend
    whose `gather` raises, or delivers other bytes to rank 0 only, moves EVERY rank to the all_gather
    form -- and the run's gather then uses it and still arrives complete and in rank order; if that
    fails too the records travel over a gloo group on the host (64 bytes per detection)."""
    import torch.multiprocessing as mp
    assert parallel.selftest_counts(1) == [3] and parallel.selftest_counts(8)[-1] == 0
    assert len(set(parallel.selftest_counts(8)[:-1])) > 1          # uneven
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selftest_worker, args=(r, world, port, break_gather, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, method, out = ret.get(timeout=180)
        got[rank] = (method, out)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert {m for m, _ in got.values()} == {want}
    counts = [4, 0, 9][:world]
    assert list(got[0][1][:, 0]) == [r + 1 for r in range(world) for _ in range(counts[r])]
    assert all(got[r][1].shape == (0, 64) for r in range(1, world))


def test_rank_env_is_the_same_on_every_launch_route(monkeypatch):
    """The driver's `torch.distributed.run ... bench.py --gpus N` never went through
    relaunch_under_torchrun(): rank_env() is what both routes apply, before torch / HIP initialise.
    A value the caller exported wins."""
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    env = {}
    assert parallel.rank_env(env) == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"} and env == parallel.RANK_ENV
    assert parallel.rank_env() == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert parallel.rank_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}) == {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}
    # bench.py and the CLI call it before they import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("parallel.rank_env()") < main.index("import torch")
    cli = open(os.path.join(root, "thrifty_amd", "detect_cli.py")).read()
    body = cli[cli.index("def detector_cli("):]
    assert body.index("parallel.rank_env()") < body.index("parallel.sharded_env(")


def test_populators_are_sized_from_the_hosts_cpus_per_rank(monkeypatch):
    monkeypatch.setattr(parallel, "cpu_budget", lambda: 16)
    assert [parallel.populate_threads(w) for w in (1, 2, 4, 8)] == [3, 3, 1, 1]
    monkeypatch.setattr(parallel, "cpu_budget", lambda: 256)
    assert [parallel.populate_threads(w) for w in (1, 8)] == [3, 3]
    monkeypatch.setattr(parallel, "cpu_budget", lambda: 2)
    assert parallel.populate_threads(1) == 1
    assert parallel.cpu_budget.__name__ == "<lambda>"       # (the real one reads affinity + cgroup cpu.max)


# ---------------------------------------------------------------------------------------------
# `thrifty detect --gpus N` host logic: reader sharding and the sharded run (gloo stands in for
# RCCL; the engine is replaced by a deterministic record maker -- no GPU here)
# ---------------------------------------------------------------------------------------------
def _card_file(tmp_path, n, block_len=64, comments=True):
    from thrifty_amd import block_data
    rng = np.random.default_rng(5)
    path = tmp_path / "rx.card"
    with open(path, "w") as f:
        if comments:
            f.write("# header line\nUsing Volk machine: avx2\n")
        for i in range(n):
            if comments and i % 5 == 3:
                f.write("# note\n\n")
            f.write(block_data.card_line(100.0 + 0.25 * i, 10 + i, rng.integers(0, 256, 2 * block_len, dtype=np.uint8)))
    return path


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("n", [0, 1, 7, 40])
def test_card_stream_shards_partition_the_file_in_order(tmp_path, n, world):
    from thrifty_amd import block_data
    path = _card_file(tmp_path, n)
    with open(path, "rb") as f:
        full = block_data.CardStream(f, 64).next_batch(10 ** 6)
    want = [] if full is None else full[1].tolist()
    got, sizes = [], []
    for r in range(world):
        with open(path, "rb") as f:
            cs = block_data.CardStream(f, 64).shard(r, world)
            mine = []
            while True:
                b = cs.next_batch(3)
                if b is None:
                    break
                stamps, idxs, text, offs = b
                for ts, i, o in zip(stamps, idxs, offs):        # payloads really are this block's
                    assert ts == 100.0 + 0.25 * (int(i) - 10)
                mine.extend(idxs.tolist())
            sizes.append(len(mine))
            got.extend(mine)
    assert got == want
    if n >= 8 * world:
        assert max(sizes) - min(sizes) <= 2, sizes


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_raw_stream_shards_partition_the_file_in_order(tmp_path, world):
    from thrifty_amd import block_data
    size, hist = 64, 24          # lead-in: ceil(24 / 40) = 1 block
    data = np.random.default_rng(6).integers(0, 256, 2 * (size - hist) * 23 + 10, dtype=np.uint8)
    path = tmp_path / "rx.bin"
    data.tofile(path)

    def collect(rs):
        out = []
        while True:
            b = rs.next_batch(4)
            if b is None:
                return out
            kind, stamps, idxs, payload = b
            step, carry = 2 * (size - hist), 2 * hist
            for k, i in enumerate(idxs.tolist()):
                if kind == "u8":
                    blk = bytes(payload[k * step:k * step + carry + step])
                    out.append((i, "u8", blk))
                else:
                    out.append((i, "c64", payload[k].tobytes()))

    with open(path, "rb") as f:
        want = collect(block_data.RawStream(f, size, hist))
    got = []
    for r in range(world):
        with open(path, "rb") as f:
            got.extend(collect(block_data.RawStream(f, size, hist).shard(r, world)))
    assert [g[0] for g in got] == list(range(23))
    assert got == want


def test_pipes_cannot_be_sharded():
    import io
    from thrifty_amd import block_data
    with pytest.raises(ValueError):
        block_data.CardStream(io.BytesIO(b""), 64).shard(1, 2)
    with pytest.raises(ValueError):
        block_data.RawStream(io.BytesIO(b""), 64, 16).shard(1, 2)


class _FakeDetections(object):
    """What run_sharded needs of a Detector: records of detected blocks for this rank's range.
    Record content is a pure function of the global block position, so the sharded result can
    be compared with a one-rank run."""

    new_len, rxid, _offset_type = 48, 3, float

    def __init__(self, lo, hi, fail_at=None):
        self.lo, self.hi, self.fail_at = lo, hi, fail_at

    @staticmethod
    def records(positions):
        from thrifty_amd import _native
        r = np.zeros(len(positions), dtype=_native.RECORD_DTYPE)
        p = np.asarray(positions, dtype=np.int64)
        r["block_idx"], r["flags"], r["carrier_bin"] = p + 1000, 3, 10 + p % 90
        r["corr_sample"], r["corr_offset"] = 5 + p % 7, 0.1 * np.sin(p)
        r["carrier_offset"] = np.cos(p) / 3
        r["corr_energy"], r["corr_noise"] = 100.0 + p, 1.0 + (p % 3)
        r["carrier_energy"], r["carrier_noise"] = 50.0 + p / 7.0, 0.5
        return r

    def iter_detected_records(self):
        for s in range(self.lo, self.hi, 5):
            pos = [p for p in range(s, min(s + 5, self.hi)) if p % 3 != 1]      # some blocks undetected
            if self.fail_at is not None and any(p >= self.fail_at for p in range(s, min(s + 5, self.hi))):
                pos = [p for p in pos if p < self.fail_at]
                if pos:
                    yield 0.5 * np.asarray(pos, dtype=np.float64), self.records(pos)
                raise IndexError("index 65 is out of bounds for axis 0 with size 64")
            if pos:
                yield 0.5 * np.asarray(pos, dtype=np.float64), self.records(pos)


def _sharded_worker(rank, world, port, n, fail_at, out_path, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    lo, hi = parallel.shard_range(n, rank, world)
    det = _FakeDetections(lo, hi, fail_at if fail_at is not None and lo <= fail_at < hi else None)
    out = open(out_path, "w") if rank == 0 else None
    try:
        parallel.run_sharded(det, rank, world, 0, out, backend="gloo")
        ret.put((rank, None))
    except IndexError as exc:
        ret.put((rank, str(exc)))
    finally:
        if out is not None:
            out.close()


@pytest.mark.parametrize("fail_at", [None, 13])
def test_run_sharded_two_ranks_writes_one_ordered_toad(tmp_path, fail_at):
    import torch.multiprocessing as mp
    from thrifty_amd import toads_data
    n, world = 23, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    out_path = str(tmp_path / "rx.toad")
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n, fail_at, out_path, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    results = dict(ret.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # what ONE process over the whole range writes (and where it would have died)
    stop = n if fail_at is None else fail_at
    pos = [p for p in range(stop) if p % 3 != 1]
    want = toads_data.toad_lines(_FakeDetections.records(pos), 0.5 * np.asarray(pos, dtype=np.float64),
                                 _FakeDetections.new_len, rxid=_FakeDetections.rxid)
    got = open(out_path).read().split("\n")
    assert got[-1] == "" and got[:-1] == want
    if fail_at is None:
        assert results == {0: None, 1: None}
    else:       # every rank raises, like the single-process loop
        assert "out of bounds" in results[1] and "IndexError" in results[0], results


class _FakeMultiDetections(_FakeDetections):
    """Multi-template flavour: T records per detected block, ordered [block][template], the
    template id in `template_id` (what MultiTemplateDetector.iter_detected_records yields)."""

    _multi, T = True, 4

    @classmethod
    def records(cls, positions):
        base = _FakeDetections.records(np.repeat(np.asarray(positions, dtype=np.int64), cls.T))
        base["template_id"] = np.tile(np.arange(cls.T), len(positions))
        base["corr_energy"] += base["template_id"]
        keep = (base["block_idx"] + base["template_id"]) % 5 != 0     # some templates undetected
        return base[keep], keep

    def iter_detected_records(self):
        for s in range(self.lo, self.hi, 5):
            pos = [p for p in range(s, min(s + 5, self.hi)) if p % 3 != 1]
            if pos:
                recs, keep = self.records(pos)
                yield np.repeat(0.5 * np.asarray(pos, dtype=np.float64), self.T)[keep], recs


def _sharded_multi_worker(rank, world, port, n, out_path, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    lo, hi = parallel.shard_range(n, rank, world)
    out = open(out_path, "w") if rank == 0 else None
    try:
        parallel.run_sharded(_FakeMultiDetections(lo, hi), rank, world, 0, out, backend="gloo")
        ret.put((rank, None))
    finally:
        if out is not None:
            out.close()


def test_run_sharded_multi_template_keeps_block_template_order_and_txid(tmp_path):
    """BASELINE configs[4] host side: T records per block cross the gather ordered
    [block][template]; rank 0 writes them with the template id as txid (the reference's
    serialize() puts txid after rxid, toads_data.py:57-61)."""
    import torch.multiprocessing as mp
    from thrifty_amd import toads_data
    n, world = 23, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    out_path = str(tmp_path / "rx.toads")
    procs = [ctx.Process(target=_sharded_multi_worker, args=(r, world, port, n, out_path, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    results = dict(ret.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert results == {0: None, 1: None}
    pos = [p for p in range(n) if p % 3 != 1]
    recs, keep = _FakeMultiDetections.records(pos)
    stamps = np.repeat(0.5 * np.asarray(pos, dtype=np.float64), _FakeMultiDetections.T)[keep]
    want = []
    for r, ts in zip(recs, stamps):      # one DetectionResult per record, serialised the reference's way
        res = toads_data.DetectionResult(
            float(ts), int(r["block_idx"]), 48 * int(r["block_idx"]) + int(r["corr_sample"]) + float(r["corr_offset"]),
            toads_data.CarrierSyncInfo(int(r["carrier_bin"]), float(r["carrier_offset"]),
                                       np.float32(r["carrier_energy"]), np.float32(r["carrier_noise"])),
            toads_data.CorrDetectionInfo(int(r["corr_sample"]), float(r["corr_offset"]),
                                         float(r["corr_energy"]), float(r["corr_noise"])),
            rxid=3, txid=int(r["template_id"]))
        want.append(res.serialize())
    got = open(out_path).read().split("\n")
    assert got[-1] == "" and got[:-1] == want
    keys = [(int(g.split()[3]), int(g.split()[1])) for g in got[:-1]]     # (block, txid)
    assert keys == sorted(keys)


def test_sharded_env_needs_the_marker(monkeypatch):
    """RANK / WORLD_SIZE inherited from an unrelated launcher do not make a plain `thrifty
    detect` a rank of a sharded run; the re-launched children carry THRIFTY_SHARDED=1."""
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.delenv("THRIFTY_SHARDED", raising=False)
    assert parallel.sharded_env() == (0, None, 0)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    assert parallel.sharded_env(4) == (0, None, 0)          # not torchrun's: `--gpus 4` re-launches
    monkeypatch.setenv("THRIFTY_SHARDED", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert parallel.sharded_env() == (1, 4, 1)


def test_a_rank_that_torchrun_started_directly_never_relaunches(monkeypatch):
    """`torchrun --nproc-per-node 4 -m thrifty_amd.detect --gpus 4 ...`: every child sees RANK /
    WORLD_SIZE and torchrun's run id but no THRIFTY_SHARDED.  It IS a rank (re-launching from
    inside it would start 4 x 4 processes and four writers of one file); a world size that
    contradicts --gpus is refused, and without --gpus the environment is ignored."""
    monkeypatch.delenv("THRIFTY_SHARDED", raising=False)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    monkeypatch.setenv("RANK", "2")
    monkeypatch.setenv("LOCAL_RANK", "2")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert parallel.sharded_env(4) == (2, 4, 2)
    assert parallel.sharded_env(1) == (0, None, 0)
    with pytest.raises(SystemExit) as exc:
        parallel.sharded_env(8)
    assert "4 ranks" in str(exc.value)


def test_gpus_without_an_output_file_is_refused():
    from thrifty_amd.detect import Detector, detector_cli
    with pytest.raises(SystemExit) as exc:
        detector_cli(Detector, argv=["rx.card", "--gpus", "2", "--quiet"])
    assert "output file" in str(exc.value)
