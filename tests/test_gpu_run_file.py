"""GPU tests of thr_run_card / thr_run_stream -- the whole `thrifty detect <file> --quiet -o <toad>`
loop inside the library (reference detect.py:197-223: reader -> Detector.detect -> `if detected:
print(result.serialize())`): byte-identical `.toad` to the batched Python loop it replaces
(`Detector.iter_toad_text`), which the other suites pin to the reference's goldens; the reference's
error behaviour (IndexError block, malformed line, invalid payload) with everything before the
offending input written; the record sink the sharded CLI uses; and the input-window release rule
for overlapping raw-stream chunks."""
import io
import os

import numpy as np
import pytest

from thrifty_amd import _native as F
from thrifty_amd import block_data, synth
from thrifty_amd.block_data import CardStream, RawStream
from thrifty_amd.detect import Detector, DetectorSettings, MultiTemplateDetector

from test_gpu_detector_api import assert_toad_close, card_text, settings_of
from test_gpu_stream import burst_stream

pytestmark = pytest.mark.gpu


def data_lines(text):
    return [ln for ln in text.strip().split("\n")
            if ln.strip() and not ln.startswith(("#", "Using Volk machine:", "linux;"))]


def python_loop_text(settings, reader, cls=Detector, **kw):
    return b"".join(cls(settings, reader, **kw).iter_toad_text())


def library_loop(settings, path, make_reader, cls=Detector, **kw):
    """write_toad() of a Detector over the regular file `path` -> (bytes written, stats)"""
    out = path + ".toad"
    with open(path, "rb") as f, open(out, "wb") as o:
        det = cls(settings, make_reader(f), **kw)
        assert det._library_loop_ready()
        stats = det.write_toad(o)
        assert not det._pin and det._exhausted           # window closed, nothing left
    return open(out, "rb").read(), stats


@pytest.mark.parametrize("name,batch", [("c2", 5), ("c2", 64), ("c1", 3), ("small", 4), ("c2_straddle", 1000)])
def test_card_file_through_the_library_loop_equals_the_python_loop_and_the_golden(golden, tmp_path, name, batch):
    g = golden(name)
    n = int(g["block_len"])
    text = str(g["card_text"]) if "card_text" in g.files else card_text(g)
    bad = np.flatnonzero(g["index_error"]) if "index_error" in g.files else []
    lines = data_lines(text)
    if len(bad):            # (the IndexError block has its own test below)
        lines = [ln for i, ln in enumerate(lines) if i not in set(bad.tolist())]
    # the lines card_reader skips (block_data.py:120-127), CRLF ends, no newline at the very end
    text = "# fastcard header\n\nUsing Volk machine: avx2\n" + lines[0] + "\r\n" + "\n".join(lines[1:])
    path = str(tmp_path / "rx.card")
    open(path, "wb").write(text.encode())
    st = settings_of(g)
    got, stats = library_loop(st, path, lambda f: CardStream(f, n), rxid=3, batch_size=batch)
    # (a rank on a node short of CPUs: one populator, waits that nap instead of polling -- same bytes)
    lean, _ = library_loop(st, path, lambda f: CardStream(f, n), rxid=3, batch_size=batch, populate_threads=1,
                           low_cpu=True)
    assert lean == got
    want = python_loop_text(st, CardStream(io.BytesIO(text.encode()), n), rxid=3, batch_size=batch)
    assert got == want and got.count(b"\n") == stats["detections"] > 0
    assert stats["blocks"] == len(lines) and stats["calls"][-1]["batches"] == -(-len(lines) // batch)
    if not len(bad) and "toad" in g.files and "card_text" not in g.files:
        ref = [" ".join(["3"] + ln.split()[1:]) for ln in str(g["toad"]).strip().split("\n")]
        assert_toad_close(got.decode().strip().split("\n"), "\n".join(ref))


def test_four_templates_txid_column_and_order(golden, tmp_path):
    gs = [golden("c5_tx%d" % i) for i in range(4)]
    st = settings_of(gs[0], np.stack([g["template"] for g in gs]))
    text = card_text(gs[0])
    path = str(tmp_path / "rx.card")
    open(path, "wb").write(text.encode())
    got, stats = library_loop(st, path, lambda f: CardStream(f, 16384), cls=MultiTemplateDetector, rxid=0,
                              batch_size=5)
    want = python_loop_text(st, CardStream(io.BytesIO(text.encode()), 16384), cls=MultiTemplateDetector,
                            rxid=0, batch_size=5)
    assert got == want and stats["detections"] >= 12
    cols = [ln.split() for ln in got.decode().strip().split("\n")]
    keys = [(int(c[3]), int(c[1])) for c in cols]                  # (block, txid)
    assert keys == sorted(keys)                                     # [block][template] order


@pytest.mark.parametrize("interp", ["parabolic", "none", "cosine"])
def test_preshift_detector_files_take_the_library_loop_too(golden, tmp_path, interp):
    """The variants are engine handles like any other: PreshiftDetector's float32 carrier offset
    (printed widened), interpolator `none`'s int 0 and cosine's early `return 0` come out of
    thr_run_card's formatter exactly as out of the Python loop."""
    from thrifty_amd.experimental.detect_preshift import PreshiftDetector
    g = golden("c2")
    st = settings_of(g)
    text = card_text(g)
    if interp == "cosine":      # some blocks with a stronger bin just below the window: the int-0 branch
        rng = np.random.default_rng(3)
        tpl, n = np.asarray(g["template"], dtype=np.float64), 16384
        extra = []
        for k, car in enumerate((6.15, 6.24, 6.41)):
            p = int(rng.integers(1537, 13825))
            z = rng.normal(0, 0.02, n) + 1j * rng.normal(0, 0.02, n) + 0.08 * np.exp(2j * np.pi * car * np.arange(n) / n)
            z[p:p + len(tpl)] += 0.3 * (tpl + 1) / 2 * np.exp(2j * np.pi * car * (np.arange(len(tpl)) + p) / n)
            extra.append(block_data.card_line(2000.0 + k, 500 + k, synth.quantise_iq(z)))
        text += "".join(extra)
    path = str(tmp_path / "rx.card")
    open(path, "wb").write(text.encode())
    got, stats = library_loop(st, path, lambda f: CardStream(f, 16384), cls=PreshiftDetector, rxid=2, num=21,
                              interpolator=interp, batch_size=6)
    want = python_loop_text(st, CardStream(io.BytesIO(text.encode()), 16384), cls=PreshiftDetector, rxid=2,
                            num=21, interpolator=interp, batch_size=6)
    assert got == want and stats["detections"] >= 18
    offsets = [ln.split()[9] for ln in got.decode().strip().split("\n")]
    if interp == "none":
        assert set(offsets) == {"0"}
    elif interp == "cosine":
        assert offsets.count("0") == 2 and all("." in o for o in offsets if o != "0")
    else:
        assert all("." in o and float(np.float32(float(o))) == float(o) for o in offsets)   # widened float32


@pytest.mark.parametrize("n,h,bits,sps,batch", [(16384, 4096, 10, 1.0, 4), (16384, 4920, 10, 1.0, 64),
                                                (65536, 4098, 11, 2.0, 3), (4096, 1024, 9, 1.0, 7)])
def test_raw_file_through_the_library_loop_equals_the_python_loop(tmp_path, n, h, bits, sps, batch):
    """`thrifty detect --raw rx.bin`: the zero-history lead-in blocks through the complex64 path,
    everything behind them in ONE thr_run_stream call -- the Python loop's text but for the
    timestamps (wall clock of the batch, block_data.py:86-98)."""
    tpl = synth.gold_template(bits, 2, sps)
    raw, _ = burst_stream(np.random.default_rng(n + h + 1), n, h, tpl, 23, (20.5, 33.3, 44.1, 63.7, 77.0, 91.2))
    path = str(tmp_path / "rx.bin")
    open(path, "wb").write(raw.tobytes())
    st = DetectorSettings(n, h, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
    got, stats = library_loop(st, path, lambda f: RawStream(f, n, h), rxid=1, batch_size=batch)
    want = python_loop_text(st, RawStream(io.BytesIO(raw.tobytes()), n, h), rxid=1, batch_size=batch)
    strip = lambda text: [" ".join(ln.split()[:1] + ln.split()[2:]) for ln in text.decode().strip().split("\n")]
    assert strip(got) == strip(want) and len(strip(got)) >= 5
    assert stats["blocks"] == 23 and stats["detections"] == len(strip(got))
    now = [float(ln.split()[1]) for ln in got.decode().strip().split("\n")]
    assert all(abs(t - now[0]) < 60 for t in now) and now[0] > 1.6e9


def test_index_error_block_ends_the_run_where_the_reference_raises(golden, tmp_path):
    """carrier_sync.py:187 indexes fft_mag[peak + 3] unwrapped: the reference's loop dies on that
    block with everything before it written.  Same here, same message as Detector.detect()."""
    g = golden("c2_straddle")
    bad = int(np.flatnonzero(g["index_error"])[0])
    assert 0 < bad < len(g["blocks"]) - 1
    path = str(tmp_path / "rx.card")
    open(path, "wb").write(card_text(g).encode())
    st = settings_of(g)
    with pytest.raises(IndexError) as one:
        Detector(st, None).detect(0.0, 0, g["blocks"][bad])
    for batch in (2, 1000):
        out = str(tmp_path / "rx.toad")
        with open(path, "rb") as f, open(out, "wb") as o:
            det = Detector(st, CardStream(f, 16384), rxid=0, batch_size=batch)
            with pytest.raises(IndexError) as exc:
                det.write_toad(o)
                assert str(exc.value) == str(one.value) and not det._pin
        want = b""
        try:
            for text in Detector(st, CardStream(io.BytesIO(card_text(g).encode()), 16384), rxid=0,
                                 batch_size=batch).iter_toad_text():
                want += text
        except IndexError:
            pass
        got = open(out, "rb").read()
        assert got == want
        blocks = [int(ln.split()[2]) for ln in got.decode().strip().split("\n") if ln]
        assert all(b < int(g["block_idx"][bad]) for b in blocks)


def test_bad_input_in_the_middle_everything_before_it_is_written(golden, tmp_path):
    g = golden("c2")
    lines = data_lines(card_text(g))
    st = settings_of(g)
    good = python_loop_text(st, CardStream(io.BytesIO(("\n".join(lines[:6]) + "\n").encode()), 16384), rxid=0)
    # (1) a malformed line (payload too short): the host framing refuses it -> ValueError, like
    # CardStream; (2) an invalid base64 character: the device decode flags it -> NativeError
    pay = lines[6].split(" ")[2]
    malformed = "\n".join(lines[:6] + [lines[6][:-8]] + lines[7:]) + "\n"
    invalid = "\n".join(lines[:6] + [" ".join(lines[6].split(" ")[:2]) + " " + pay[:50] + "!" + pay[51:]]
                        + lines[7:]) + "\n"
    # (batches of 3: lines 0-5 precede the bad one's batch; batches of 4: the malformed line is the
    # third of its batch -- the framing hands out the two lines before it first, thr_frame_card)
    cases = [(malformed, ValueError, "payload", 3), (malformed, ValueError, "payload", 4),
             (invalid, F.NativeError, "base64", 3)]
    for text, exc_type, word, batch in cases:
        path, out = str(tmp_path / "rx.card"), str(tmp_path / "rx.toad")
        open(path, "wb").write(text.encode())
        with open(path, "rb") as f, open(out, "wb") as o:
            det = Detector(st, CardStream(f, 16384), rxid=0, batch_size=batch)
            with pytest.raises(exc_type, match=word):
                det.write_toad(o)
        assert open(out, "rb").read() == good
        # the Python loop over the same text stops at the same place
        seen = b""
        with pytest.raises(exc_type, match=word):
            for chunk in Detector(st, CardStream(io.BytesIO(text.encode()), 16384), rxid=0,
                                  batch_size=batch).iter_toad_text():
                seen += chunk
        assert seen == good
        # and the handle is still usable afterwards
        rec = det._engine.detect(g["blocks"][:2], g["block_idx"][:2])[:, 0]
        assert rec[0]["corr_sample"] == g["sample"][0]


def test_record_sink_equals_the_python_record_iteration(golden, tmp_path):
    g = golden("c2")
    path = str(tmp_path / "rx.card")
    open(path, "wb").write(card_text(g).encode())
    st = settings_of(g)
    with open(path, "rb") as f:
        got = Detector(st, CardStream(f, 16384), rxid=0, batch_size=5).detected_records()
    chunks = list(Detector(st, CardStream(io.BytesIO(card_text(g).encode()), 16384), rxid=0,
                           batch_size=5).iter_detected_records())
    stamps = np.concatenate([c[0] for c in chunks])
    want = np.concatenate([c[1] for c in chunks]).copy()
    want["reserved"] = stamps.view(np.uint64)
    assert got.tobytes() == want.tobytes() and len(got) == int(g["det"].sum())


def test_run_arguments_are_checked(golden, tmp_path):
    g = golden("c2")
    eng = Detector(settings_of(g), None, batch_size=8)._engine
    text = card_text(g).encode()
    with pytest.raises(F.NativeError, match="neither an output descriptor nor a record array"):
        eng.run_card(text)
    with pytest.raises(F.NativeError, match="batch_blocks 9 exceeds"):
        eng.run_card(text, rec_out=np.zeros(64, dtype=F.RECORD_DTYPE), batch_blocks=9)
    with pytest.raises(F.NativeError, match="more detections than rec_capacity"):
        eng.run_card(text, rec_out=np.zeros(2, dtype=F.RECORD_DTYPE))
    st = eng.run_card(b"# nothing but a comment\n\n", rec_out=np.zeros(2, dtype=F.RECORD_DTYPE))
    assert st["blocks"] == 0 and st["detections"] == 0 and not st["index_error"]


def test_cli_quiet_output_runs_inside_the_library(golden, tmp_path, monkeypatch):
    """`thrifty detect --quiet rx.card -o rx.toad` takes the library loop (and says so through the
    statistics write_toad returns); `-a` appends through the same descriptor."""
    from thrifty_amd.detect import detector_cli
    g = golden("c2")
    np.save(tmp_path / "template.npy", g["template"])
    (tmp_path / "detector.cfg").write_text(
        "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
        "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
        "template: %s\n" % (tmp_path / "template.npy"))
    (tmp_path / "rx.card").write_text(card_text(g))
    seen = []
    real = Detector.write_toad
    monkeypatch.setattr(Detector, "write_toad", lambda self, out: seen.append(real(self, out)))
    base = [str(tmp_path / "rx.card"), "--quiet", "-c", str(tmp_path / "detector.cfg")]
    detector_cli(Detector, argv=base + ["-o", str(tmp_path / "rx.toad")])
    detector_cli(Detector, argv=base + ["-a", str(tmp_path / "rx.toad")])
    assert len(seen) == 2 and all(s is not None and s["detections"] == int(g["det"].sum()) for s in seen)
    lines = (tmp_path / "rx.toad").read_text().strip().split("\n")
    half = len(lines) // 2
    assert lines[:half] == lines[half:]
    assert_toad_close(lines[:half], g["toad"])


def test_the_cli_closes_its_engine_and_a_script_that_never_does_still_exits(golden, tmp_path):
    """`detector_cli` destroys the engine when its loop ends (threads joined, pages unlocked) instead
    of leaving it to the interpreter's shutdown; and a script that drops out with an engine and an
    input window still open -- the window's threads inside the HIP runtime -- exits cleanly: an
    atexit hook of thrifty_amd._native closes what is left before the runtime's own destructors."""
    import subprocess
    import sys
    from thrifty_amd.detect import detector_cli
    g = golden("c2")
    np.save(tmp_path / "template.npy", g["template"])
    (tmp_path / "detector.cfg").write_text(
        "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
        "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
        "template: %s\n" % (tmp_path / "template.npy"))
    (tmp_path / "rx.card").write_text(card_text(g))
    made = []

    class Watched(Detector):
        def __init__(self, *a, **kw):
            super(Watched, self).__init__(*a, **kw)
            made.append(self)

    for extra in (["--quiet"], []):
        detector_cli(Watched, argv=[str(tmp_path / "rx.card"), "-c", str(tmp_path / "detector.cfg"),
                                    "-o", str(tmp_path / "rx.toad")] + extra)
    assert len(made) == 2 and all(d._engine._h is None for d in made)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import mmap, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from thrifty_amd import _native as F\n"
        "from thrifty_amd.detect import Detector, DetectorSettings\n"
        "from thrifty_amd.block_data import CardStream\n"
        "tpl = np.load(%r)\n"
        "st = DetectorSettings(16384, 4096, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))\n"
        "f = open(%r, 'rb')\n"
        "det = Detector(st, CardStream(f, 16384), rxid=0)\n"
        "assert det._pin\n"
        "n = sum(len(b) for b in det.iter_toad_lines())\n"
        "eng = F.Engine(16384, 4096, tpl, (0, 15, 0), (7, 110), (0, 15, 0))\n"
        "mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)\n"
        "eng.input_window(memoryview(mm))\n"
        "print('lines', n, len(F._live_engines))\n"
    ) % (root, str(tmp_path / "template.npy"), str(tmp_path / "rx.card"))
    for _ in range(3):
        res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stderr[-2000:]
        assert res.stdout.split()[:2] == ["lines", str(int(g["det"].sum()))] and int(res.stdout.split()[2]) >= 2


def test_window_is_never_released_past_the_start_of_an_open_chunk():
    """Chunks of a raw stream overlap by 2 * history bytes.  With three of them in flight, collecting
    the oldest must not release (let the worker unlock) input-window segments that the next chunk's
    copy still reads: the released mark stays at or below the start of every open chunk, also when a
    segment boundary falls inside the overlap (64 KiB segments put one into nearly every overlap)."""
    # (history 3/4 of the block: consecutive 10-block chunks start 20 KiB apart and share 6 KiB, so a
    # 64 KiB boundary falls into the shared bytes of about every third pair of chunks)
    n, h = 4096, 3072
    tpl = synth.gold_template(9, 2, 1.0)
    nblk = 2000
    raw, _ = burst_stream(np.random.default_rng(11), n, h, tpl, nblk, (20.5, 44.1, 63.7, 91.2, 33.0, 50.5))
    room = np.empty(len(raw) + 4096, dtype=np.uint8)        # a page-aligned copy: the same segment grid every run
    off = -room.ctypes.data % 4096
    stream = room[off:off + len(raw)]
    stream[:] = raw
    step, blk = 2 * (n - h), 2 * n
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=10)
    lead = -(-h // (n - h))                       # blocks that still reach before the stream's first byte
    tail = stream[lead * step - 2 * h:]           # blocks lead .. : block i starts (i - lead) * step into `tail`
    want = eng.detect_stream(tail, first_block_idx=lead)                          # no window: pageable copies
    eng.input_window(stream, segment_bytes=1 << 16)
    base = stream.ctypes.data & ~4095
    total = (len(tail) - blk) // step + 1
    chunks = [(s, min(10, total - s)) for s in range(0, total, 10)]
    open_, got, inside = [], [], 0
    pending = list(chunks)
    while pending or open_:
        while pending and len(open_) < F.MAX_IN_FLIGHT:
            s, nb = pending.pop(0)
            view = tail[s * step:(s + nb - 1) * step + blk]
            open_.append((eng.submit_stream(view, first_block_idx=lead + s), view.ctypes.data - base, len(view)))
        ticket, start, size = open_.pop(0)
        got.append(eng.collect(ticket))
        released = eng.debug_window()[0]
        for _, o_start, _ in open_:
            assert released <= o_start, (released, o_start)
            # (the case the rule exists for: the next chunk starts in the segment BELOW this one's end)
            inside += (o_start >> 16) < ((start + size - 1) >> 16)
    eng.input_window(None)
    got = np.concatenate(got)
    assert inside >= 10
    assert got.tobytes() == want.tobytes() and len(got) == total
