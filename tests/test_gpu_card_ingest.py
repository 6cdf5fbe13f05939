"""GPU tests of the on-device .card ingest: base64 decode kernel +
CardStream batch reader, against the host decode path and the reference goldens."""
import io
import os

import numpy as np
import pytest

from thrifty_amd import _native as F
from thrifty_amd import block_data
from thrifty_amd.block_data import CardStream
from thrifty_amd.detect import Detector

from test_gpu_parity import check_against_golden, engine_for
from test_gpu_detector_api import assert_toad_close, card_text, settings_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c2", "small"])
def test_device_decode_equals_host_decode(golden, name):
    g = golden(name)
    text = card_text(g).encode()
    cs = CardStream(io.BytesIO(text), int(g["block_len"]))
    stamps, idxs, buf, offs = cs.next_batch(1000)
    assert np.array_equal(idxs, g["block_idx"]) and stamps[0] == 1000.0
    eng = engine_for(g, max_batch=5)            # forces several chunks per call
    rec_card = eng.detect_card(buf, offs, idxs)[:, 0]
    rec_host = eng.detect(g["blocks"], g["block_idx"])[:, 0]
    for f in rec_host.dtype.names:
        assert np.array_equal(rec_card[f], rec_host[f]), f      # bit-identical
    check_against_golden(rec_card, g)


def test_detector_over_card_stream_matches_reference_toad(golden):
    g = golden("small")
    det = Detector(settings_of(g), CardStream(io.BytesIO(str(g["card_text"]).encode()), 4096,
                                               chunk_bytes=50000), rxid=3, batch_size=4)
    lines = [res.serialize() for detected, res in det if detected]
    assert_toad_close(lines, g["card_toad"])


def test_bad_payloads_are_rejected(golden):
    g = golden("c2")
    eng = engine_for(g)
    line = block_data.card_line(1.0, 7, g["blocks"][0])
    bad = line.replace("A", "!", 1) if "A" in line.split(" ")[2] else line[:40] + "!" + line[41:]
    cs = CardStream(io.BytesIO(bad.encode()), 16384)
    _, idxs, buf, offs = cs.next_batch(4)
    with pytest.raises(F.NativeError, match="base64"):
        eng.detect_card(buf, offs, idxs)
    # padding in the middle of a payload is invalid too
    pay = line.split(" ")[2]
    mid = " ".join(line.split(" ")[:2]) + " " + pay[:100] + "=" + pay[101:]
    _, idxs, buf, offs = CardStream(io.BytesIO(mid.encode()), 16384).next_batch(4)
    with pytest.raises(F.NativeError):
        eng.detect_card(buf, offs, idxs)
    # wrong payload length is caught by the host-side framing
    with pytest.raises(ValueError, match="payload"):
        CardStream(io.BytesIO((line[:-9] + "\n").encode()), 16384).next_batch(4)
    # and the engine still works afterwards
    _, idxs, buf, offs = CardStream(io.BytesIO(line.encode()), 16384).next_batch(4)
    rec = eng.detect_card(buf, offs, idxs)[:, 0]
    assert rec[0]["block_idx"] == 7 and rec[0]["corr_sample"] == g["sample"][0]


def test_mapped_input_file_is_page_locked_and_gives_the_same_records(golden, tmp_path):
    """`thrifty detect rx.card` / `--raw rx.bin` on a regular file: the mapping is the engine's
    input window (thr_input_window: a library thread page-locks it ahead of the chunk copies, which
    are then asynchronous DMA from the page cache), closed when the last batch is out; records are
    those of the pageable path, byte for byte."""
    g = golden("c2")
    n = int(g["block_len"])
    text = (card_text(g) * 40).encode()          # 640 lines: several engine batches of 64
    path = tmp_path / "rx.card"
    path.write_bytes(text)
    outs = []
    for pin in (True, False):
        with open(path, "rb") as f:
            det = Detector(settings_of(g), CardStream(f, n), rxid=0, batch_size=64, pin_input=pin)
            assert det._pin == pin
            outs.append(b"".join(det.iter_toad_text()))
            assert det._pin is False                              # window closed at the end of the run
    assert outs[0] == outs[1] and outs[0].count(b"\n") > 300
    # raw stream file
    raw = np.concatenate([g["blocks"][i][: 2 * (n - int(g["history_len"]))] for i in range(len(g["blocks"]))] * 8)
    rpath = tmp_path / "rx.bin"
    rpath.write_bytes(raw.tobytes())
    routs = []
    for pin in (True, False):
        with open(rpath, "rb") as f:
            det = Detector(settings_of(g), block_data.RawStream(f, n, int(g["history_len"])), rxid=0,
                           batch_size=16, pin_input=pin)
            assert det._pin == pin
            routs.append([(d, r.block, r.soa) for d, r in det])
    assert routs[0] == routs[1] and len(routs[0]) > 100
    # thr_host_register (whole-range form) on an ordinary buffer: best effort, refusal is harmless
    pin = F.HostPin(memoryview(bytearray(1 << 20)), limit=16)
    assert not pin.ok and "limit" in pin.why
    buf = np.zeros(1 << 22, dtype=np.uint8)
    pin = F.HostPin(buf)
    assert pin.ok, pin.why
    pin.close()
    assert not pin.ok


def test_input_window_on_a_file_larger_than_the_lock_ahead_distance(golden, tmp_path):
    """1.2 GB of .card text: more than the 1 GiB the worker may lock ahead, so segments are
    locked, read and unlocked on the go; a second pass over the same Engine re-opens the window;
    reading a range behind the window falls back to a pageable copy.  Same records every way."""
    g = golden("c2")
    n = int(g["block_len"])
    lines = card_text(g).encode().split(b"\n")[:-1]
    nlines = len(lines) * (30000 // len(lines))
    path = tmp_path / "big.card"
    with open(path, "wb") as f:
        for i in range(nlines):
            f.write(lines[i % len(lines)] + b"\n")
    st = settings_of(g)
    with open(path, "rb") as f:
        det = Detector(st, CardStream(f, n), rxid=0)
        assert det._pin and path.stat().st_size > (900 << 20)
        text = b"".join(det.iter_toad_text())
    per = b"".join(Detector(st, CardStream(io.BytesIO(b"\n".join(lines) + b"\n"), n), rxid=0).iter_toad_text())
    assert per.count(b"\n") > 0 and text.count(b"\n") == per.count(b"\n") * (nlines // len(lines))
    # an Engine with a window: a range behind the read position and a range outside it both work
    import mmap
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        cs = CardStream(io.BytesIO(b"\n".join(lines[i % len(lines)] for i in range(70)) + b"\n"), n)
        stamps, idxs, buf, offs = cs.next_batch(64)
        eng = engine_for(g, max_batch=64)
        ref = eng.detect_card(buf, offs, idxs)
        eng.input_window(mm)
        cs2 = CardStream(f, n)
        b2 = cs2.next_batch(64)
        inside = eng.detect_card(b2[2], b2[3], b2[1])          # from the window
        outside = eng.detect_card(buf, offs, idxs)              # a private copy: not in the window
        again = eng.detect_card(b2[2], b2[3], b2[1])           # the same range once more
        eng.input_window(None)
        assert inside.tobytes() == ref.tobytes() == outside.tobytes() == again.tobytes()
        eng.close()
        del cs2, b2
        mm.close()


def test_device_ingest_equals_the_reference_native_card_reader(golden, tmp_path):
    """The whole `.card` front end of the engine -- thr_frame_card framing, base64 decode on the
    device -- against the reference's own native reader (fastcard/card_reader.c + lib/base64.c,
    oracle/_ref): detection records from the text == records from the bytes the reference
    decoded, byte for byte."""
    from oracle import ref_readers
    if not ref_readers.available():
        # absent only where it cannot be built; required wherever the checkout is, or the caller says so
        from test_ref_readers import ref_required
        assert not ref_required(), \
            "oracle/_ref/libfastcard_readers.so is missing: run __graft_entry__.build() / make -C oracle"
        pytest.skip("oracle/_ref/libfastcard_readers.so not built (needs the reference checkout)")
    g = golden("c2")
    n = int(g["block_len"])
    path = tmp_path / "rx.card"
    path.write_text("# capture\n" + card_text(g))
    ref, rc = ref_readers.read_blocks(str(path), n, int(g["history_len"]), card=True)
    assert rc == 1 and len(ref) == len(g["blocks"])
    ref_bytes = np.stack([r[3] for r in ref])
    assert np.array_equal(ref_bytes, g["blocks"])
    eng = engine_for(g, max_batch=8)
    with open(path, "rb") as f:
        cs = CardStream(f, n)
        stamps, idxs, buf, offs = cs.next_batch(1000)
        rec_text = eng.detect_card(buf, offs, idxs)
    assert [int(i) for i in idxs] == [r[2] for r in ref]
    assert all(abs(ts - (r[0] + r[1] * 1e-6)) < 1e-9 for ts, r in zip(stamps, ref))
    rec_ref = eng.detect(ref_bytes, np.asarray([r[2] for r in ref], dtype=np.int64))
    assert rec_text.tobytes() == rec_ref.tobytes()
