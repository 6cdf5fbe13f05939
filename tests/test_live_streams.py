"""Host-side stream behaviour that the reference has and a batching reader can lose:
live pipes deliver per record (reference card_reader / block_reader return per line / per
block, block_data.py:70-131), timestamps are per block, errors surface at their block."""
import io
import os
import threading
import time

import numpy as np
import pytest

from thrifty_amd import _native, block_data, detect, fastdet, toads_data


def _card_text(n, block_len, rng):
    lines, raws = [], []
    for i in range(n):
        raw = rng.integers(0, 256, 2 * block_len, dtype=np.uint8)
        raws.append(raw)
        lines.append(block_data.card_line(1000.0 + i, i, raw).encode())
    return lines, raws


def test_card_stream_on_a_live_pipe_returns_per_arrival():
    """Three .card lines written 0.3 s apart through an OS pipe wrapped in a BufferedReader
    (what sys.stdin.buffer is): each must come out before the next one is written."""
    rng = np.random.default_rng(0)
    lines, _ = _card_text(3, 256, rng)
    rfd, wfd = os.pipe()
    reader = os.fdopen(rfd, "rb")          # BufferedReader: readinto() would block for a full chunk
    arrivals = []

    def produce():
        with os.fdopen(wfd, "wb", buffering=0) as w:
            for ln in lines:
                w.write(ln)
                time.sleep(0.3)

    # (the framing routine lives in the engine library: load it BEFORE the clock starts -- the first
    # dlopen of a 12 MB library can take longer than the 0.25 s this test allows the first line)
    block_data.CardStream(io.BytesIO(lines[0]), 256).next_batch(4)
    th = threading.Thread(target=produce)
    t0 = time.perf_counter()
    th.start()
    cs = block_data.CardStream(reader, 256)
    got = []
    while True:
        batch = cs.next_batch(1024)
        if batch is None:
            break
        arrivals.append(time.perf_counter() - t0)
        got.extend(batch[1].tolist())
    th.join()
    assert got == [0, 1, 2]
    assert len(arrivals) == 3, arrivals                 # one batch per arrival, not one at EOF
    assert arrivals[0] < 0.25 and arrivals[1] < 0.55, arrivals


def test_raw_stream_on_a_live_pipe_stamps_each_block_at_arrival():
    size, hist = 256, 64
    new = size - hist
    rng = np.random.default_rng(1)
    chunks = [rng.integers(0, 256, 2 * new, dtype=np.uint8).tobytes() for _ in range(5)]
    rfd, wfd = os.pipe()
    reader = os.fdopen(rfd, "rb")

    def produce():
        with os.fdopen(wfd, "wb", buffering=0) as w:
            w.write(chunks[0])            # lead-in block (zero history)
            time.sleep(0.05)
            for c in chunks[1:]:
                w.write(c)
                time.sleep(0.2)

    th = threading.Thread(target=produce)
    th.start()
    rs = block_data.RawStream(reader, size, hist)
    stamps, idxs = [], []
    while True:
        batch = rs.next_batch(1024)
        if batch is None:
            break
        stamps.extend(batch[1])
        idxs.extend(batch[2].tolist())
    th.join()
    assert idxs == [0, 1, 2, 3, 4]
    gaps = np.diff(stamps[1:])
    assert np.all(gaps > 0.1), stamps       # not one time.time() for the whole stream


def test_raw_stream_one_read_many_blocks_still_get_stamps():
    size, hist = 256, 64
    new = size - hist
    data = np.random.default_rng(2).integers(0, 256, 2 * new * 7, dtype=np.uint8).tobytes()
    import io
    rs = block_data.RawStream(io.BytesIO(data), size, hist)
    n = 0
    while True:
        b = rs.next_batch(4)
        if b is None:
            break
        assert len(b[1]) == len(b[2])
        n += len(b[2])
    assert n == 7


def test_card_line_microsecond_carry():
    raw = np.zeros(8, dtype=np.uint8)
    assert block_data.card_line(12.9999996, 3, raw).startswith("13.000000 3 ")
    assert block_data.card_line(12.25, 3, raw).startswith("12.250000 3 ")
    res = toads_data.DetectionResult(
        12.9999996, 5, 100.5, toads_data.CarrierSyncInfo(10, 0.25, 3.0, 1.0),
        toads_data.CorrDetectionInfo(7, 0.5, 9.0, 2.0), 1)
    assert fastdet.fastdet_line(res).split(" ")[1] == "13.000000"


class _FakeEngine(object):
    """Engine stand-in for host-logic tests (no GPU here): hands back prepared records."""

    def __init__(self, recs):
        self.recs = recs

    def detect(self, arr, idx):
        out = self.recs[:len(idx)].copy()
        out["block_idx"] = idx
        self.recs = self.recs[len(idx):]
        return out.reshape(-1, 1)

    def submit(self, arr, idx):          # the asynchronous pair the iterator drives
        self.open += 1
        return self.detect(arr, idx)

    def collect(self, ticket):
        self.open -= 1
        return ticket

    open = 0


def _bare_detector(recs, blocks, batch_size, max_wait=float("inf"), max_fill=float("inf"), known_not_live=True):
    d = detect.Detector.__new__(detect.Detector)
    d.settings = detect.DetectorSettings(64, 16, 8, (0, 15, 0), (0, -1), np.ones(8), (0, 15, 0))
    d._card = d._raw = None
    d.blocks = iter(blocks)
    d.rxid, d.yield_data, d.batch_size, d.new_len = 7, False, batch_size, 48
    d._engine = _FakeEngine(recs)
    from collections import deque
    d._ready, d._exhausted, d.only_detections = deque(), False, False
    d._ahead, d._depth, d.max_wait, d.max_fill = deque(), 1, max_wait, max_fill
    d._read_error, d._known_not_live = None, known_not_live
    return d


def test_index_error_surfaces_at_its_block_after_earlier_results():
    """Reference: the per-block loop emits blocks 0..k-1, then raises at block k
    (carrier_sync.py:187).  A batch must not swallow the earlier results."""
    recs = np.zeros(6, dtype=_native.RECORD_DTYPE)
    recs["flags"] = [3, 0, 3, 4, 3, 3]
    recs["carrier_bin"] = [10, 11, 12, 62, 13, 14]
    recs["corr_sample"] = 5
    blocks = [(float(i), i, np.zeros(64, dtype=np.complex64)) for i in range(6)]
    d = _bare_detector(recs, blocks, batch_size=6)
    out = []
    try:
        for det, res in d:
            out.append((det, res.block))
        raised = None
    except IndexError as exc:
        raised = str(exc)
    assert out == [(True, 0), (False, 1), (True, 2)]
    assert raised == "index 65 is out of bounds for axis 0 with size 64"
    # and the iterator is finished afterwards, like a crashed loop
    assert list(d) == []
    assert d._engine.open == 0        # the batch that was in flight behind the error was collected


def test_batches_are_submitted_one_ahead_and_all_collected():
    """File-like sources: batch i + 1 is submitted before batch i is collected (the device works
    while the host formats), results still come out in input order, no ticket stays open."""
    recs = np.zeros(10, dtype=_native.RECORD_DTYPE)
    recs["flags"] = 3
    recs["corr_sample"] = np.arange(10)
    blocks = [(float(i), i, np.zeros(64, dtype=np.complex64)) for i in range(10)]
    d = _bare_detector(recs, blocks, batch_size=3)
    first = next(d)
    assert first[1].block == 0 and d._engine.open == 1      # batch 1 is in flight behind batch 0
    assert [r.block for _, r in d] == list(range(1, 10))
    assert d._engine.open == 0


def test_slow_source_is_not_held_for_a_full_batch():
    recs = np.zeros(4, dtype=_native.RECORD_DTYPE)

    def slow():
        for i in range(4):
            time.sleep(0.05)
            yield float(i), i, np.zeros(64, dtype=np.complex64)

    d = _bare_detector(recs, slow(), batch_size=1024, max_wait=0.002)   # what a `.live` source gets
    t0 = time.perf_counter()
    first = next(d)
    assert time.perf_counter() - t0 < 0.15      # not 4 x 0.05 s (the whole source)
    assert first[1].block == 0
    assert [r.block for _, r in d] == [1, 2, 3]


def test_format_toad_is_serialize_byte_for_byte():
    """thr_format_toad (host routine of the engine library) == DetectionResult.serialize() ==
    the reference's format string (toads_data.py:47-61): shortest-repr floats, %.6f / %.8f,
    np.float32 fields printed widened, ids in front."""
    rng = np.random.default_rng(3)
    n = 4000
    recs = np.zeros(n, dtype=_native.RECORD_DTYPE)
    recs["block_idx"] = rng.integers(0, 1 << 34, n)
    recs["template_id"] = rng.integers(0, 8, n)
    recs["carrier_bin"] = rng.integers(0, 16384, n)
    recs["corr_sample"] = rng.integers(0, 16384, n)
    recs["corr_offset"] = rng.uniform(-0.6, 0.6, n)
    recs["carrier_offset"] = rng.standard_normal(n) * 10.0 ** rng.integers(-9, 3, n)
    for f in ("carrier_energy", "carrier_noise", "corr_energy", "corr_noise"):
        recs[f] = (rng.standard_normal(n) * 10.0 ** rng.integers(-8, 18, n)).astype(np.float32)
    special = [0.0, -0.0, 1e16, 1e15, 1e-4, 1e-5, 123456789012345678.0, 1.0, -1.5, 5e-324, 0.1, 2.5e-7]
    recs["carrier_offset"][:len(special)] = special
    recs["corr_energy"][:4] = [0.0, 1.0, 16777216.0, 1e-30]
    stamps = rng.uniform(0, 2e9, n)
    stamps[:4] = [0.0, 0.0000005, 1234567890.1234565, 0.9999995]      # rounding ties / carries
    for rxid, multi, f32 in ((None, False, False), (-1, False, False), (7, True, False), (2, False, True)):
        text = _native.format_toad(recs, stamps, 12288, rxid=rxid, with_txid=multi, carrier_offset_f32=f32)
        want = []
        for r, ts in zip(recs, stamps):
            off = np.float32(r["carrier_offset"]) if f32 else float(r["carrier_offset"])
            res = toads_data.DetectionResult(
                float(ts), int(r["block_idx"]),
                12288 * int(r["block_idx"]) + int(r["corr_sample"]) + float(r["corr_offset"]),
                toads_data.CarrierSyncInfo(int(r["carrier_bin"]), off, np.float32(r["carrier_energy"]),
                                           np.float32(r["carrier_noise"])),
                toads_data.CorrDetectionInfo(int(r["corr_sample"]), float(r["corr_offset"]),
                                             float(r["corr_energy"]), float(r["corr_noise"])),
                rxid=rxid, txid=int(r["template_id"]) if multi else None)
            want.append(res.serialize())
        assert text.decode().split("\n") == want + [""]
    assert _native.format_toad(recs[:0], stamps[:0], 1) == b""


class _FakeMultiEngine(_FakeEngine):
    def detect(self, arr, idx):
        t = 3
        out = self.recs[:len(idx) * t].copy().reshape(len(idx), t)
        out["block_idx"] = np.asarray(idx)[:, None]
        out["template_id"] = np.arange(t)[None, :]
        self.recs = self.recs[len(idx) * t:]
        return out


def test_multi_template_iteration_groups_per_block_with_txid():
    recs = np.zeros(12, dtype=_native.RECORD_DTYPE)
    recs["flags"] = [3, 1, 3, 0, 0, 0, 3, 3, 3, 1, 1, 3]      # block 1: no carrier at all
    recs["corr_sample"] = np.arange(12)
    blocks = [(float(i), 10 + i, np.zeros(64, dtype=np.complex64)) for i in range(4)]
    d = _bare_detector(recs, blocks, batch_size=3)
    d.__class__ = detect.MultiTemplateDetector
    d.n_templates = 3
    d._engine = _FakeMultiEngine(recs)
    out = list(d)
    assert [len(per_tx) for per_tx in out] == [3, 3, 3, 3]
    assert [[(det, res.block, res.txid) for det, res in per_tx] for per_tx in out] == [
        [(True, 10, 0), (False, 10, 1), (True, 10, 2)], [(False, 11, 0), (False, 11, 1), (False, 11, 2)],
        [(True, 12, 0), (True, 12, 1), (True, 12, 2)], [(False, 13, 0), (False, 13, 1), (True, 13, 2)]]
    # the flat record iterator: detections only, [block][template] order, stamps repeated
    d2 = _bare_detector(recs, blocks, batch_size=3)
    d2.__class__, d2.n_templates, d2._engine = detect.MultiTemplateDetector, 3, _FakeMultiEngine(recs)
    flat = [(float(ts), int(r["block_idx"]), int(r["template_id"]))
            for stamps, rr in d2.iter_detected_records() for ts, r in zip(stamps, rr)]
    assert flat == [(0.0, 10, 0), (0.0, 10, 2), (2.0, 12, 0), (2.0, 12, 1), (2.0, 12, 2), (3.0, 13, 2)]
    # only_detections: a block's detected templates, blocks without any are skipped
    d3 = _bare_detector(recs, blocks, batch_size=4)
    d3.__class__, d3.n_templates, d3._engine = detect.MultiTemplateDetector, 3, _FakeMultiEngine(recs)
    d3.only_detections = True
    assert [[(res.block, res.txid) for _, res in per_tx] for per_tx in d3] == [
        [(10, 0), (10, 2)], [(12, 0), (12, 1), (12, 2)], [(13, 2)]]


def test_is_live_and_reader_readiness(tmp_path):
    """A pipe is live (blocks arrive as they are captured), a file or an in-memory stream is not;
    a batch reader says whether its next batch would have to be WAITED for -- the Detector reads
    ahead of the batch it hands out only when it would not."""
    import io
    path = tmp_path / "x.bin"
    path.write_bytes(bytes(4096))
    with open(path, "rb") as f:
        assert not block_data.is_live(f)
        assert block_data.RawStream(f, 64, 16).ready() and not block_data.RawStream(f, 64, 16).live
    assert not block_data.is_live(io.BytesIO(b"abc"))
    rfd, wfd = os.pipe()
    with os.fdopen(rfd, "rb") as r, os.fdopen(wfd, "wb", buffering=0) as w:
        assert block_data.is_live(r)
        assert block_data.card_reader(r).live and block_data.block_reader(r, 64, 16).live
        cs = block_data.CardStream(r, 64)
        assert cs.live and not cs.mapped and not cs.ready()       # nothing has arrived
        raw = np.arange(128, dtype=np.uint8)
        w.write(block_data.card_line(1.0, 0, raw).encode())
        assert cs.ready()                                         # readable now
        assert cs.next_batch(8)[1].tolist() == [0]
        assert not cs.ready()                                     # consumed, pipe empty again


def test_a_live_source_is_not_read_ahead_of():
    """Reference behaviour on a pipe: block i's result is out before block i + 1 exists
    (block_data.py:101-131 yields per line).  The batching Detector over a CardStream must neither
    wait for a full batch nor submit the NEXT batch before handing out this one."""
    recs = np.zeros(3, dtype=_native.RECORD_DTYPE)
    recs["flags"] = 3
    rng = np.random.default_rng(2)
    lines, _ = _card_text(3, 64, rng)
    rfd, wfd = os.pipe()
    reader = os.fdopen(rfd, "rb")
    w = os.fdopen(wfd, "wb", buffering=0)
    d = _bare_detector(recs, (), batch_size=1024)    # (the batch reader below is the source)
    d._card = block_data.CardStream(reader, 64)
    eng = d._engine
    eng.submit_card = lambda text, offs, idxs: eng.submit(None, idxs)
    eng.inputs_consumed = lambda ticket: None
    w.write(lines[0])
    t0 = time.perf_counter()
    first = next(d)                      # must not block on line 1 (nobody has written it yet)
    assert time.perf_counter() - t0 < 0.5 and first[1].block == 0 and eng.open == 0
    w.write(lines[1])
    w.write(lines[2])
    w.close()
    assert [r.block for _, r in d] == [1, 2]
    assert eng.open == 0


def test_a_source_that_does_not_say_whether_it_is_live_gets_bounded_latency_and_no_read_ahead():
    """Any user generator / filter / socket reader has no `.live`.  It must neither be waited on
    for a whole batch (batch_size blocks at a receiver's 200 blocks/s = seconds) nor be read
    ahead of: filling a batch stops after `max_fill`, and the next batch is pulled only when the
    caller asks for it."""
    settings = detect.DetectorSettings(64, 16, 8, (0, 15, 0), (0, -1), np.ones(8), (0, 15, 0))
    pulled = []

    def slow():
        for i in range(6):
            time.sleep(0.03)
            pulled.append(i)
            yield float(i), i, np.zeros(64, dtype=np.complex64)

    # what Detector.__init__ derives for such a source (no GPU here: the engine is faked)
    assert getattr(slow(), "live", None) is None
    recs = np.zeros(6, dtype=_native.RECORD_DTYPE)
    d = _bare_detector(recs, slow(), batch_size=1024, max_fill=detect._UNKNOWN_FILL_S, known_not_live=False)
    t0 = time.perf_counter()
    first = next(d)
    assert time.perf_counter() - t0 < 0.15 and first[1].block == 0
    assert len(pulled) <= 3 and d._in_flight is None          # nothing pulled behind the caller's back
    assert [r.block for _, r in d] == [1, 2, 3, 4, 5]
    # a reader that declares itself file-backed fills whole batches and is read ahead of
    class Listed(list):
        live = False
    assert detect.Detector.__init__.__defaults__ is not None
    d2 = _bare_detector(recs, [(float(i), i, np.zeros(64, dtype=np.complex64)) for i in range(6)], batch_size=2)
    next(d2)
    assert d2._in_flight is not None


def test_defaults_follow_the_live_attribute():
    """max_wait / max_fill as Detector.__init__ derives them from `.live` (True / False / absent)."""
    import inspect
    src = inspect.getsource(detect.Detector.__init__)
    assert "_UNKNOWN_FILL_S if live is None" in src and "_SLOW_SOURCE_S if live else" in src
    assert detect._UNKNOWN_FILL_S <= 0.1 and detect._SLOW_SOURCE_S <= 0.01


def test_in_memory_sequences_count_as_not_live():
    """A list / tuple / ndarray of blocks has no `.live` attribute but never has to be waited for:
    it is read ahead of and fills whole batches (the docstring said so; the code treated it as
    'does not say' -- 50 ms batches, no read-ahead)."""
    import inspect
    src = inspect.getsource(detect.Detector.__init__)
    assert "isinstance(blocks, (list, tuple, np.ndarray))" in src
    assert src.index("isinstance(blocks, (list, tuple, np.ndarray))") < src.index("self._known_not_live =")


def test_an_index_error_leaves_no_stale_read_error_and_no_locked_window():
    """After the IndexError block ended the iteration, an error the read-ahead had parked (it
    belongs to blocks BEHIND the one that ended the loop) must not surface on the next call, and a
    page-locked input must be let go: next() -> StopIteration, window closed."""
    recs = np.zeros(8, dtype=_native.RECORD_DTYPE)
    recs["flags"] = [3, 3, 4, 3, 3, 3, 3, 3]
    recs["carrier_bin"] = 62

    def source():
        for i in range(6):
            yield float(i), i, np.zeros(64, dtype=np.complex64)
        raise ValueError("malformed .card line far behind the block that ended the loop")

    d = _bare_detector(recs, source(), batch_size=3)
    closed = []
    d._engine.input_window = lambda buf=None, **kw: closed.append(buf)
    d._pin = True
    out = []
    with pytest.raises(IndexError):
        for det, res in d:
            out.append(res.block)
    assert out == [0, 1]
    with pytest.raises(StopIteration):
        next(d)                             # not the parked ValueError
    assert d._read_error is None and closed == [None] and not d._pin
    assert d._engine.open == 0


def test_an_error_in_the_read_ahead_does_not_swallow_the_batch_before_it():
    """Reference: everything before a malformed line is emitted, then the loop dies.  The batch
    that was being collected when the NEXT batch failed to read must still be handed out."""
    recs = np.zeros(8, dtype=_native.RECORD_DTYPE)
    recs["flags"] = 3

    def source():
        for i in range(5):
            yield float(i), i, np.zeros(64, dtype=np.complex64)
        raise ValueError("malformed .card line")

    d = _bare_detector(recs, source(), batch_size=3)
    out = []
    with pytest.raises(ValueError, match="malformed"):
        for det, res in d:
            out.append(res.block)
    assert out == [0, 1, 2, 3, 4]           # batch 0 AND the partial batch before the bad item
    assert d._engine.open == 0
    assert list(d) == []


def test_a_partial_record_is_not_ready():
    """A producer that does not flush at record boundaries (stdio's 4 KiB buffering) leaves half a
    line / half a block in the pipe: next_batch() would block on it, so ready() must say no."""
    raw = np.arange(128, dtype=np.uint8)
    line = block_data.card_line(1.0, 0, raw).encode()
    rfd, wfd = os.pipe()
    with os.fdopen(rfd, "rb") as r, os.fdopen(wfd, "wb", buffering=0) as w:
        cs = block_data.CardStream(r, 64)
        w.write(b"# a comment line\n" + line[:40])
        released = []
        assert not cs.ready(release=lambda: released.append(1))      # half a line has arrived
        assert released == [1]                                      # ... and was taken in, after release()
        w.write(line[40:])
        assert cs.ready()
        assert cs.next_batch(8)[1].tolist() == [0]
        assert not cs.ready()
    rfd, wfd = os.pipe()
    with os.fdopen(rfd, "rb") as r, os.fdopen(wfd, "wb", buffering=0) as w:
        # (realistic sizes: reads larger than the BufferedReader's 8 KiB go straight to the pipe;
        # smaller ones may park bytes in its internal buffer, where select() cannot see them --
        # ready() then errs on the safe side and says no)
        rs = block_data.RawStream(r, 8192, 2048)      # 12288 new bytes per block
        step = 2 * (8192 - 2048)
        w.write(bytes(step // 2))
        assert not rs.ready()                          # half a block
        w.write(bytes(step - step // 2))
        assert rs.ready()
        w.write(bytes(step * 3))
        assert rs.next_batch(8)[0] == "c64" and rs.ready()       # the lead-in block; three more wait
        got = rs.next_batch(8)
        assert got[0] == "u8" and len(got[2]) == 3 and not rs.ready()
