"""GPU tests of the raw-stream framing on the device (second half):
`thr_detect_stream*` reads overlapping blocks in place from the receiver's u8 byte stream
instead of the host re-copying each block's history (reference block_data.py:70-98,
fastcard raw_reader.c:15-46)."""
import io

import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import block_data, synth
from thrifty_amd.detect import Detector, DetectorSettings

pytestmark = pytest.mark.gpu


def burst_stream(rng, n, h, tpl, nblk, carriers):
    """Continuous noise with one burst per entry of `carriers`, quantised to u8 I/Q."""
    new = n - h
    x = rng.normal(0, 0.02, new * nblk) + 1j * rng.normal(0, 0.02, new * nblk)
    ook = 0.3 * (np.asarray(tpl, float) + 1) / 2
    starts = []
    for j, car in enumerate(carriers):
        start = int((j + 0.5) * new * nblk / len(carriers))
        k = np.arange(len(tpl))
        x[start:start + len(tpl)] += ook * np.exp(2j * np.pi * car * (k + start) / n)
        starts.append(start)
    return synth.quantise_iq(x), starts


def framed(raw, n, h, first, count):
    """Explicit host framing of blocks first .. first+count-1 (all-u8 region only)."""
    step = 2 * (n - h)
    return np.stack([raw[i * step - 2 * h: i * step - 2 * h + 2 * n] for i in range(first, first + count)])


@pytest.mark.parametrize("n,h,bits,sps", [
    (16384, 4096, 10, 1.0),     # fast path, BASELINE C2 geometry
    (16384, 4920, 10, 1.0),     # example-config history (stride not a multiple of 64 bytes)
    (32768, 4096, 11, 1.0),     # long path, R0 = 2 (three overlap-save sections on strided blocks)
    (65536, 4098, 11, 2.0),     # BASELINE C3's template, five sections, stride not a multiple of 16 bytes
    (4096, 1024, 9, 1.0),       # short-block LDS path (4 blocks per workgroup)
    (8192, 2050, 10, 1.0),      # short blocks, block starts only 4-byte aligned (stride 12284 B)
    (2048, 502, 8, 1.0),        # 8 blocks per workgroup, stride 3092 B
    (1024, 254, 7, 1.0),        # 16 blocks per workgroup
])
def test_stream_framing_equals_host_framing(n, h, bits, sps):
    tpl = synth.gold_template(bits, 2, sps)
    rng = np.random.default_rng(n + h)
    nblk = 11
    raw, _ = burst_stream(rng, n, h, tpl, nblk, (20.5, 44.1, 63.7, 91.2))
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=4)   # forces 3 chunks
    # stream bytes from the start of block 1's history onwards == blocks 1 .. nblk-1
    step = 2 * (n - h)
    tail = raw[step - 2 * h:]
    rec_s = eng.detect_stream(tail, first_block_idx=1)[:, 0]
    assert len(rec_s) == nblk - 1
    rec_h = eng.detect(framed(raw, n, h, 1, nblk - 1), np.arange(1, nblk))[:, 0]
    assert rec_s.tobytes() == rec_h.tobytes()            # same kernels, same bytes -> identical records
    assert (rec_s["flags"] & F.FLAG_CORR).sum() >= 3


def test_stream_device_entry_point():
    import torch
    n, h = 16384, 4096
    tpl = synth.gold_template(10, 2, 1.0)
    raw, _ = burst_stream(np.random.default_rng(5), n, h, tpl, 9, (30.2, 77.7))
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=16)
    step = 2 * (n - h)
    d_stream = torch.from_numpy(raw[step - 2 * h:].copy()).cuda()
    nb = (d_stream.numel() - 2 * n) // step + 1
    d_idx = torch.arange(1, nb + 1, dtype=torch.int64, device="cuda")
    d_out = torch.zeros(nb * F.RECORD_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.detect_stream_device(d_stream.data_ptr(), nb, d_out.data_ptr(), d_idx.data_ptr())
    eng.sync()
    rec_d = d_out.cpu().numpy().view(F.RECORD_DTYPE)
    rec_h = eng.detect(framed(raw, n, h, 1, nb), np.arange(1, nb + 1))[:, 0]
    assert rec_d.tobytes() == rec_h.tobytes()


@pytest.mark.parametrize("n,h", [(16384, 4096), (16384, 12000)])   # 1 and 3 lead-in blocks
def test_detector_over_rawstream_matches_oracle(n, h):
    """Detector(RawStream) == oracle over block_reader framing, lead-in blocks included."""
    tpl = synth.gold_template(10, 2, 1.0)
    nblk = 7 if h == 4096 else 14
    raw, starts = burst_stream(np.random.default_rng(n - h), n, h, tpl, nblk, (25.5, 58.3, 88.8))
    settings = DetectorSettings(block_len=n, history_len=h, carrier_len=len(tpl), carrier_thresh=(0, 15, 0),
                                carrier_window=(7, 110), template=tpl, corr_thresh=(0, 15, 0))
    src = block_data.RawStream(io.BytesIO(raw.tobytes()), n, h)
    det = Detector(settings, src, rxid=4, batch_size=4)
    assert det._raw is src
    got = list(det)
    orc = onp.OracleDetector(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0))
    want = [orc.detect_block(idx, np.asarray(blk))[0]
            for _, idx, blk in block_data.block_reader(io.BytesIO(raw.tobytes()), n, h)]
    assert len(got) == len(want) == nblk
    hits = 0
    for (detected, res), w in zip(got, want):
        assert detected == w.detected
        assert res.carrier_info.bin == w.carrier.bin
        if w.detected:
            hits += 1
            assert res.corr_info.sample == w.corr.sample
            np.testing.assert_allclose(res.corr_info.offset, w.corr.offset, atol=1e-4)
            np.testing.assert_allclose(res.corr_info.energy, w.corr.energy, rtol=1e-4)
            np.testing.assert_allclose(res.soa, w.soa, atol=2e-4)
    # a burst lying in a block's overlap zone (outside its unique window) can additionally trip
    # the threshold with a correlation sidelobe -- in the reference just the same (parity above)
    assert hits >= 3
    soas = np.array([r.soa for d, r in got if d])
    for s0 in starts:
        assert np.min(np.abs(soas - (s0 + h))) < 1.0


def test_odd_stride_is_refused_and_detector_falls_back_to_host_framing():
    n, h = 4096, 1023          # new = 3073 samples: block starts are not 4-byte aligned
    tpl = synth.gold_template(9, 2, 1.0)
    raw, _ = burst_stream(np.random.default_rng(2), n, h, tpl, 5, (40.4,))
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=4)
    with pytest.raises(F.NativeError, match="even block_len - history_len"):
        eng.detect_stream(raw)
    settings = DetectorSettings(block_len=n, history_len=h, carrier_len=len(tpl), carrier_thresh=(0, 15, 0),
                                carrier_window=(7, 110), template=tpl, corr_thresh=(0, 15, 0))
    src = block_data.RawStream(io.BytesIO(raw.tobytes()), n, h)
    det = Detector(settings, src, batch_size=4)
    assert det._raw is None
    got = list(det)
    assert len(got) == 5 and sum(d for d, _ in got) == 1


def test_short_stream_yields_no_blocks():
    n, h = 16384, 4096
    eng = F.Engine(n, h, synth.gold_template(10, 2, 1.0), (0, 15, 0), (7, 110), (0, 15, 0), max_batch=4)
    assert eng.detect_stream(np.zeros(2 * n - 2, np.uint8)).shape == (0, 1)
