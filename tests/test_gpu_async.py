"""Asynchronous host boundary of the C ABI (include/thrifty_hip.h: thr_submit* / thr_collect /
thr_inputs_consumed / thr_poll): tickets interleave, collect in any order, the records are the
synchronous entry points' byte for byte; and the Detector iterator that rides on it."""
import io

import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import block_data, synth
from thrifty_amd.detect import Detector, DetectorSettings, MultiTemplateDetector

pytestmark = pytest.mark.gpu

N, H = 16384, 4096


@pytest.fixture(scope="module")
def case():
    tpl = synth.gold_template(10, 2).astype(np.float64)
    win = onp.unique_window(N, H, len(tpl))
    blocks, _ = synth.synth_blocks(np.random.default_rng(31), 96, N, tpl, win, signal_frac=0.7)
    return tpl, blocks


def test_two_tickets_interleave_and_equal_the_synchronous_call(case):
    tpl, blocks = case
    eng = F.Engine(N, H, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=40)
    want = eng.detect(blocks, np.arange(96) + 5)
    a = eng.submit(blocks[:40], np.arange(40) + 5)
    b = eng.submit(blocks[40:72], np.arange(40, 72) + 5)       # second batch in flight behind the first
    assert a.id != b.id and a.id and b.id
    eng.inputs_consumed(b)                                      # (its inputs may now be overwritten)
    c = eng.submit(blocks[72:], np.arange(72, 96) + 5)
    with pytest.raises(F.NativeError, match="in flight"):       # THR_MAX_IN_FLIGHT = 3
        eng.submit(blocks[:1])
    with pytest.raises(F.NativeError, match="not collected"):   # synchronous calls wait their turn
        eng.detect(blocks[:1])
    rb = eng.collect(b)                                         # any order
    ra = eng.collect(a)
    assert eng.poll(c) in (True, False)
    eng.sync()
    assert eng.poll(c) is True                                  # finished: collect would not block
    rc = eng.collect(c)
    with pytest.raises(F.NativeError, match="not open"):
        eng.collect(c)                                          # exactly once
    with pytest.raises(F.NativeError, match="not open"):
        eng.poll(c)
    got = np.concatenate([ra, rb, rc])
    assert got.tobytes() == want.tobytes()
    # the handle is back to normal
    assert eng.detect(blocks[:3], np.arange(3) + 5).tobytes() == want[:3].tobytes()
    # an empty batch: ticket 0, nothing to collect
    t = eng.submit(blocks[:0])
    assert t.id == 0 and eng.collect(t).shape == (0, 1)


def test_submit_card_and_stream_equal_the_synchronous_forms(case):
    tpl, blocks = case
    eng = F.Engine(N, H, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=64)
    text = "".join(block_data.card_line(10.0 + i, 100 + i, blocks[i]) for i in range(50)).encode()
    ts, idx, off, _ = F.frame_card(text, 0, len(text), N, True, 1000)
    want = eng.detect_card(text, off, idx)
    t1 = eng.submit_card(text, off[:30], idx[:30])
    t2 = eng.submit_card(text, off[30:], idx[30:])
    got = np.concatenate([eng.collect(t1), eng.collect(t2)])
    assert got.tobytes() == want.tobytes() == eng.detect(blocks[:50], idx).tobytes()
    # raw stream: overlapping blocks framed on the device
    stream = np.random.default_rng(5).integers(100, 156, 2 * ((N - H) * 40 + H), dtype=np.uint8)
    want = eng.detect_stream(stream, 7)
    t = eng.submit_stream(stream, 7)
    assert eng.collect(t).tobytes() == want.tobytes() and len(want) == 40


def test_detector_iteration_rides_on_the_tickets(case, tmp_path):
    """Detector over a file: batches of 16 -> several submits with one in flight ahead; the
    results equal one synchronous batch, in input order."""
    tpl, blocks = case
    st = DetectorSettings(N, H, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
    path = tmp_path / "rx.card"
    path.write_text("".join(block_data.card_line(10.0 + i, i, blocks[i]) for i in range(96)))
    whole = Detector(st, batch_size=96).detect_batch([(10.0 + i, i, blocks[i]) for i in range(96)])
    with open(path, "rb") as f:
        got = list(Detector(st, block_data.CardStream(f, N), batch_size=16))
    assert [(d, r.block, r.soa) for d, r in got] == [(d, r.block, r.soa) for d, r in whole]
    with open(path, "rb") as f:
        text = b"".join(Detector(st, block_data.CardStream(f, N), rxid=4, batch_size=16).iter_toad_text())
    assert text.decode().split("\n")[:-1] == [r.serialize() for d, r in
                                               Detector(st, rxid=4, batch_size=96).detect_batch(
                                                   [(10.0 + i, i, blocks[i]) for i in range(96)]) if d]
    # a direct detect() between two next() calls (a batch is in flight then) still works
    with open(path, "rb") as f:
        det = Detector(st, block_data.CardStream(f, N), batch_size=16)
        first = next(det)
        assert det._in_flight is not None
        d, r = det.detect(10.0 + 40, 40, blocks[40])
        assert (d, r.block, r.soa) == (whole[40][0], 40, whole[40][1].soa)
        rest = list(det)
        assert [(x[0], x[1].block) for x in [first] + rest] == [(x[0], x[1].block) for x in whole]
    # a pipe-like source (no mmap): same answer through the refilled buffer
    got2 = list(Detector(st, block_data.CardStream(io.BytesIO(path.read_bytes()), N, chunk_bytes=1 << 20),
                         batch_size=16))
    assert [(d, r.block, r.soa) for d, r in got2] == [(d, r.block, r.soa) for d, r in whole]


def test_multi_template_detector_text_and_order(case):
    tpl, blocks = case
    tpls = np.stack([synth.gold_template(10, 2 + i) for i in range(4)]).astype(np.float64)
    st = DetectorSettings(N, H, tpls.shape[1], (0, 15, 0), (7, 110), tpls, (0, 15, 0))
    items = [(10.0 + i, i, blocks[i]) for i in range(40)]
    multi = MultiTemplateDetector(st, iter(items), rxid=9, batch_size=16)
    per_block = list(multi)
    assert len(per_block) == 40 and all(len(p) == 4 for p in per_block)
    single = Detector(DetectorSettings(N, H, tpls.shape[1], (0, 15, 0), (7, 110), tpls[0], (0, 15, 0)),
                      rxid=9, batch_size=64).detect_batch(items)
    for (d1, r1), per_tx in zip(single, per_block):       # template 0 of the multi run == the single run
        d0, r0 = per_tx[0]
        assert (d0, r0.block, r0.txid) == (d1, r1.block, 0)
        if d1:
            assert r0.corr_info.sample == r1.corr_info.sample
            assert abs(r0.corr_info.energy - r1.corr_info.energy) <= 2e-5 * r1.corr_info.energy
    lines = [ln for lines in MultiTemplateDetector(st, iter(items), rxid=9, batch_size=16).iter_toad_lines()
             for ln in lines]
    want = [res.serialize() for per_tx in per_block for det, res in per_tx if det]
    assert lines == want and len(want) >= 20
    assert all(ln.split()[0] == "9" and ln.split()[1] in "0123" for ln in lines)


def test_default_stream_is_ordered_with_torch_fills():
    """Engine.set_stream(0) -- the handle value of torch's default stream -- runs the engine on the
    device's legacy default stream: a torch fill queued there is ordered before the engine's
    kernels with no explicit synchronisation (it used to select the private stream)."""
    import torch
    dev = torch.device("cuda", 0)
    tpl = synth.gold_template(10, 2).astype(np.float64)
    win = onp.unique_window(N, H, len(tpl))
    blocks, _ = synth.synth_blocks(np.random.default_rng(8), 64, N, tpl, win)
    eng = F.Engine(N, H, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=64)
    want = eng.detect(blocks, np.arange(64))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    assert torch.cuda.current_stream().cuda_stream == 0
    host = torch.from_numpy(blocks).pin_memory()
    for _ in range(5):
        data = torch.empty((64, 2 * N), dtype=torch.uint8, device=dev)
        big = torch.empty((1 << 28,), dtype=torch.uint8, device=dev)
        big.fill_(1)                                   # keeps the default stream busy for a while
        data.copy_(host, non_blocking=True)            # queued behind it, still in flight when we launch
        rec = torch.zeros((64, 64), dtype=torch.uint8, device=dev)
        eng.detect_device(data.data_ptr(), F.THR_IN_U8, 64, rec.data_ptr(), None)
        got = rec.cpu().numpy().view(F.RECORD_DTYPE).reshape(64, 1)     # default-stream copy: ordered too
        assert got.tobytes() == want.tobytes()
    eng.use_own_stream()
