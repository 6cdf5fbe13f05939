"""Host-side stream behaviour that the reference has and a batching reader can lose:
live pipes deliver per record (reference card_reader / block_reader return per line / per
block, block_data.py:70-131), timestamps are per block, errors surface at their block."""
import os
import threading
import time

import numpy as np

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


def _bare_detector(recs, blocks, batch_size):
    d = detect.Detector.__new__(detect.Detector)
    d.settings = detect.DetectorSettings(64, 16, 8, (0, 15, 0), (0, -1), np.ones(8), (0, 15, 0))
    d._card = d._raw = None
    d.blocks = iter(blocks)
    d.rxid, d.yield_data, d.batch_size, d.new_len = 7, False, batch_size, 48
    d._engine = _FakeEngine(recs)
    from collections import deque
    d._ready, d._exhausted, d.only_detections = deque(), False, False
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


def test_slow_source_is_not_held_for_a_full_batch():
    recs = np.zeros(4, dtype=_native.RECORD_DTYPE)

    def slow():
        for i in range(4):
            time.sleep(0.05)
            yield float(i), i, np.zeros(64, dtype=np.complex64)

    d = _bare_detector(recs, slow(), batch_size=1024)
    t0 = time.perf_counter()
    first = next(d)
    assert time.perf_counter() - t0 < 0.15      # not 4 x 0.05 s (the whole source)
    assert first[1].block == 0
    assert [r.block for _, r in d] == [1, 2, 3]
