"""CPU tests of the host side: settings grammar, block framing, .toad text, the C-ABI
library's exports and its refusal to run without a GPU.  Known answers are the
tables of the reference's own unit tests (tests/test_setting_parsers.py:12-99,
test_settings.py, test_block_data.py:13-71) re-expressed against thrifty_amd.
"""
import argparse
import io
import os
import re
import subprocess

import numpy as np
import pytest

from thrifty_amd import _native, block_data, setting_parsers, settings, toads_data, util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- parsers
@pytest.mark.parametrize("string,expected", [
    ("100", (100.0, 100.0, False)), ("-123.4", (-123.4, -123.4, False)),
    ("100-200", (100.0, 200.0, False)), ("10e1 - 20e1", (100.0, 200.0, False)),
    ("-100-100", (-100.0, 100.0, False)), ("-200--100", (-200.0, -100.0, False)),
    ("100hz", (100.0, 100.0, True)), ("100-200 Hz", (100.0, 200.0, True)),
    ("10-20 khz", (10000.0, 20000.0, True)), ("1.2345-2.3456 KHZ", (1.2345e3, 2.3456e3, True)),
    ("433-435Mhz", (433e6, 435e6, True)), ("1.2345-2.3456 mhz", (1.2345e6, 2.3456e6, True)),
    ("0--1", (0.0, -1.0, False)), ("7 - 110", (7.0, 110.0, False))])
def test_freq_range(string, expected):
    assert setting_parsers.freq_range(string) == expected


def test_freq_range_invalid():
    with pytest.raises(ValueError):
        setting_parsers.freq_range("garbage")


def test_normalize_freq_range():
    assert setting_parsers.normalize_freq_range((7.0, 110.0, False), 146.48) == (7, 110)
    assert setting_parsers.normalize_freq_range((-81e3, -79e3, True), 2.2e6 / 8192) == (-301, -294)


@pytest.mark.parametrize("string,expected", [("1337.15", 1337.15), ("15.2M", 15200000.0),
                                             ("987k", 987000.0), ("55m", 0.055), ("164u ", 164e-6)])
def test_metric_float(string, expected):
    assert setting_parsers.metric_float(string) == expected


@pytest.mark.parametrize("string", ["garbage", "x53m", "500A"])
def test_metric_float_invalid(string):
    with pytest.raises(ValueError):
        setting_parsers.metric_float(string)


@pytest.mark.parametrize("string,expected", [
    ("0", (0.0, 0.0, 0.0)), ("10.2", (10.2, 0.0, 0.0)), ("c", (1.0, 0.0, 0.0)),
    ("11c", (11.0, 0.0, 0.0)), ("100 * constant", (100.0, 0.0, 0.0)), ("snr", (0.0, 1.0, 0.0)),
    ("5.2*snr", (0.0, 5.2, 0.0)), (" 8s ", (0.0, 8.0, 0.0)), ("stddev", (0.0, 0.0, 1.0)),
    ("2.1stddev", (0.0, 0.0, 2.1)), ("8.7*d", (0.0, 0.0, 8.7)), ("10 + 4*snr", (10.0, 4.0, 0)),
    ("40 + 3.8*snr + 2 stddev", (40.0, 3.8, 2.0)), ("1+2s+3d+4+5s+6d", (5.0, 7.0, 9.0)),
    ("c + s + d", (1.0, 1.0, 1.0)), ("15 * snr", (0.0, 15.0, 0.0))])
def test_threshold(string, expected):
    assert setting_parsers.threshold(string) == expected


@pytest.mark.parametrize("string", ["", " ", "5+", "junk", "+5*stddev", "*snr", "stddev*snr",
                                    "5 * stdde", "2 * sn", "const"])
def test_threshold_invalid(string):
    with pytest.raises(ValueError):
        setting_parsers.threshold(string)


# ---------------------------------------------------------------- settings
DEFS = {
    "foo": settings.Definition(["--foo", "-f"], float, "2e6", None),
    "bar.baz": settings.Definition(["--baz", "-b"], float, "1e6", None),
    "xyzzy": settings.Definition(["--xyzzy", "-x"], str, None, None),
}


def test_settings_defaults_config_args():
    assert settings.load(None, None, DEFS) == {"foo": 2e6, "bar.baz": 1e6}
    assert settings.load(None, io.StringIO("bar.baz:   1234.56"), DEFS)["bar.baz"] == 1234.56
    assert settings.load(None, io.BytesIO(b"bar.baz:   1234.56 # c"), DEFS)["bar.baz"] == 1234.56
    vals = settings.load({"bar.baz": "7.8", "foo": "9.0"}, io.StringIO("bar.baz: 12.34"), DEFS)
    assert vals == {"foo": 9.0, "bar.baz": 7.8}


def test_settings_errors():
    with pytest.raises(settings.ConfigSyntaxError):
        settings.load(None, io.StringIO("foobar"), DEFS)
    with pytest.raises(settings.SettingKeyError):
        settings.load(None, io.StringIO("foobar: 1"), DEFS)
    with pytest.raises(settings.SettingKeyError):
        settings.load({"foobar": "1"}, None, DEFS)


def test_settings_argparse_and_load_args(tmp_path):
    parser = argparse.ArgumentParser()
    settings.add_argparse_arguments(parser, ["foo", "bar.baz"], definitions=DEFS)
    args = vars(parser.parse_args(["-f", "12.34", "--baz=56.78"]))
    assert args["foo"] == "12.34" and args["bar.baz"] == "56.78"
    cfg = tmp_path / "thrift.cfg"
    cfg.write_text("xyzzy: xyz\nfoo: 1.2\nbar.baz: 3.6")
    parser = argparse.ArgumentParser()
    parser.add_argument("-a", dest="a")
    config, extra = settings.load_args(parser, ["xyzzy", "foo"],
                                       argv=["-a", "extra", "--foo=2.3", "-c", str(cfg)],
                                       definitions=DEFS)
    extra.pop("verbose")
    assert dict(config) == {"xyzzy": "xyz", "foo": 2.3} and dict(extra) == {"a": "extra"}


def test_example_detector_cfg_keys():
    """The reference's example/detector.cfg (values restated) parses to config #1."""
    text = ("rxid: 0\nsample_rate: 2.4M\nchip_rate: 0.999707M\ntuner_freq: 433.83M\n"
            "tuner_gain: 0.0\ncapture_skip: 20000\nblock_size: 16384\nblock_history: 4920\n"
            "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15 * snr\n"
            "template: template.npy\n")
    v = settings.load(None, io.StringIO(text))
    assert v["block_size"] == 16384 and v["block_history"] == 4920 and v["rxid"] == 0
    assert v["carrier_window"] == (7.0, 110.0, False)
    assert v["carrier_threshold"] == (0.0, 15.0, 0.0) and v["sample_rate"] == 2.4e6


# ---------------------------------------------------------------- block framing
def test_raw_complex_round_trip():
    raw = np.array([0, 0, 127, 128, 255, 255], dtype=np.uint8)
    cplx = np.array([-0.9953 - 0.9953j, -0.0031 + 0.0047j, 0.9969 + 0.9969j], dtype=np.complex64)
    np.testing.assert_allclose(block_data.raw_to_complex(raw), cplx, rtol=1e-2)
    np.testing.assert_array_equal(block_data.complex_to_raw(cplx), raw)
    every = np.arange(256, dtype=np.uint8)
    np.testing.assert_array_equal(block_data.complex_to_raw(block_data.raw_to_complex(every)), every)


def test_block_reader_history():
    blocks = list(block_data.block_reader(io.BytesIO(bytes(range(14))), 3, 1))
    assert [b[1] for b in blocks] == [0, 1, 2]
    assert [list(block_data.complex_to_raw(b[2])) for b in blocks] == [
        [0x7f, 0x7f, 0, 1, 2, 3], [2, 3, 4, 5, 6, 7], [6, 7, 8, 9, 10, 11]]
    assert blocks[0][2].raw is None          # zero history is not a u8 value
    assert list(blocks[1][2].raw) == [2, 3, 4, 5, 6, 7]
    assert list(blocks[2][2].raw) == [6, 7, 8, 9, 10, 11]


def test_card_reader_and_writer(golden):
    stream = io.StringIO("# Some comments\n# more comments\n1000.5425 10 r0+Om5==\n"
                         "Using Volk machine: avx2\n\n1000.5442 20 aaaaaa==")
    blocks = list(block_data.card_reader(stream))
    assert [b[0] for b in blocks] == [1000.5425, 1000.5442]
    assert [b[1] for b in blocks] == [10, 20]
    assert [tuple(block_data.complex_to_raw(b[2])) for b in blocks] == [(175, 79, 142, 155),
                                                                        (105, 166, 154, 105)]
    g = golden("small")
    got = list(block_data.card_reader(io.BytesIO(str(g["card_text"]).encode())))
    assert [b[1] for b in got] == list(g["block_idx"])
    for b, raw in zip(got, g["blocks"]):
        np.testing.assert_array_equal(b[2].raw, raw)
    line = block_data.card_line(got[3][0], got[3][1], got[3][2].raw)
    assert line in str(g["card_text"])


# ---------------------------------------------------------------- .toad text
def test_toad_serialize_matches_reference_text(golden):
    """Re-serialising the reference's own .toad lines is the identity."""
    g = golden("c2")
    for line in str(g["toad"]).split("\n"):
        rec = toads_data.DetectionResult.deserialize(line, with_rxid=True)
        parts = line.split()
        # rebuild with the exact float reprs the reference printed
        assert rec.rxid == 0 and rec.block == int(parts[2])
        assert rec.serialize().split()[:3] == parts[:3]
        assert float(rec.serialize().split()[3]) == float(parts[3])
    loaded = toads_data.load_toad(io.StringIO(str(g["toad"])))
    assert len(loaded) == int(g["det"].sum())
    arr = toads_data.toads_array(loaded)
    assert np.array_equal(arr["sample"], g["sample"][g["det"]])


def test_fft_bin_and_snr():
    for num in (15, 16):
        got = np.array([util.fft_bin(i, num) for i in range(num)])
        np.testing.assert_array_equal(got, np.fft.fftfreq(num, 1. / num))
    assert abs(util.snr(10.0, 1.0) - 20.0) < 1e-12


# ---------------------------------------------------------------- C ABI library
def _lib_or_skip():
    if not os.path.exists(_native.LIB_PATH):
        pytest.skip("libthriftyhip.so not built (run __graft_entry__.build())")


def test_library_exports_every_declared_symbol():
    _lib_or_skip()
    header = open(os.path.join(ROOT, "include", "thrifty_hip.h")).read()
    declared = set(re.findall(r"\b(thr_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_native.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _native.LIB_PATH]).decode()
    for sym in declared:
        assert (" T " + sym) in out, sym
    lib = _native.load_library()
    m = re.search(r"#define THR_ABI_VERSION (\d+)", header)
    assert lib.thr_abi_version() == int(m.group(1)) == _native.ABI_VERSION
    assert _native.RECORD_DTYPE.itemsize == 64


def test_no_entry_point_lets_a_cpp_exception_out():
    """"No exceptions across the ABI" (include/thrifty_hip.h): every `int thr_*` entry point with a
    body of its own is a function-try-block that ends in thr::on_exception (host memory, a thread
    the OS refuses); the one-liners only forward to such a function or read a constant."""
    csrc = os.path.join(ROOT, "thrifty_amd", "csrc")
    seen = set()
    for name in sorted(os.listdir(csrc)):
        if not name.endswith(".hip"):
            continue
        text = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r'^(?:extern "C" )?int (thr_[a-z_0-9]+)\(', text, re.M):
            sym = m.group(1)
            rest = text[m.start():]
            head = rest[:rest.index("{") + 1]
            line_end = rest.index("\n", len(head) - 1)
            if rest[:line_end].rstrip().endswith("}"):         # a one-line body
                body = rest[len(head):line_end]
                assert re.fullmatch(r"\s*return [\w:]+(\(.*\))?;\s*}", body), (sym, body)
                seen.add(sym)
                continue
            assert re.search(r"\)\s*try \{$", head), sym
            end = rest.index("\n}\n")
            assert rest[:end + 2].rstrip().endswith('return thr::on_exception("%s");\n}' % sym), sym
            seen.add(sym)
    declared = set(re.findall(r"\bint\s+(thr_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "thrifty_hip.h")).read()))
    assert declared <= seen, sorted(declared - seen)


def test_engine_fails_loudly_without_gpu():
    """No silent CPU fallback: without a HIP device construction raises."""
    _lib_or_skip()
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(_native.NativeError):
        _native.Engine(16384, 4096, np.ones(1023), (0, 15, 0), (7, 110), (0, 15, 0))


def test_bad_settings_rejected():
    _lib_or_skip()
    with pytest.raises(_native.NativeError):
        _native.Engine(16000, 4096, np.ones(1023), (0, 15, 0), (7, 110), (0, 15, 0))  # not 2^k


# ---------------------------------------------------------------- CardStream framing (host part of f1)
def test_card_stream_framing_matches_card_reader(golden):
    g = golden("small")
    text = str(g["card_text"])
    ref = list(block_data.card_reader(io.StringIO(text)))
    for chunk in (1, 20000, 1 << 20):                     # refills in the middle of lines too
        cs = block_data.CardStream(io.BytesIO(text.encode()), 4096, chunk_bytes=chunk)
        got = []
        while True:
            batch = cs.next_batch(5)
            if batch is None:
                break
            stamps, idxs, buf, offs = batch
            assert len(stamps) <= 5 and idxs.dtype == np.int64
            for ts, idx, off in zip(stamps, idxs, offs):
                import base64
                raw = np.frombuffer(base64.b64decode(bytes(buf[off:off + cs.payload_chars])), np.uint8)
                got.append((ts, int(idx), raw))
        assert [(a, b) for a, b, _ in got] == [(r[0], r[1]) for r in ref]
        assert all(np.array_equal(x[2], r[2].raw) for x, r in zip(got, ref))
    # iterable like card_reader, text-mode streams and CRLF line ends included
    crlf = text.replace("\n", "\r\n")
    it = list(block_data.CardStream(io.StringIO(crlf), 4096))
    assert len(it) == len(ref) and all(np.array_equal(a[2], b[2]) for a, b in zip(it, ref))


def test_card_stream_rejects_malformed_lines():
    with pytest.raises(ValueError):
        block_data.CardStream(io.BytesIO(b"1000.5 7 QUJD\n"), 4096).next_batch(1)     # short payload
    with pytest.raises(ValueError):
        block_data.CardStream(io.BytesIO(b"garbage-without-fields\n"), 4096).next_batch(1)
    assert block_data.CardStream(io.BytesIO(b"# only comments\n\n"), 4096).next_batch(1) is None


# ---------------------------------------------------------------- no hidden knobs in the shipped library
def test_default_build_reads_no_environment_variables():
    """A detector's output and scheduling depend on its settings and on thr_create_ex's explicit
    arguments only.  The development switches (THR_DEV_STRIDE0 reads block 0's samples for every
    block, THR_NO_PRUNE swaps the carrier kernel, the timeline stamps) exist only behind -DTHR_DEV,
    which thrifty_amd/build.py never sets: the default library holds no THR_* name to look up."""
    import subprocess
    from thrifty_amd import build
    lib = build.LIB
    names = subprocess.check_output(["strings", "-a", lib]).decode("latin-1").split("\n")
    assert [n for n in names if re.match(r"^THR_[A-Z0-9_]+$", n)] == []
    # (the library does import getenv: rocPRIM's device-radix-sort configuration inside identify.hip
    # reads its own variables; none of ours.)  In our sources every getenv sits inside #ifdef THR_DEV
    csrc = os.path.join(ROOT, "thrifty_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp")):
            continue
        depth_dev = []
        for ln in open(os.path.join(csrc, f)):
            t = ln.strip()
            if t.startswith("#if"):
                depth_dev.append(t.startswith("#ifdef THR_DEV"))
            elif t.startswith("#endif") and depth_dev:
                depth_dev.pop()
            elif "getenv(" in t and not t.startswith("//"):
                assert any(depth_dev), (f, t)
    assert "THR_DEV" not in open(os.path.join(ROOT, "thrifty_amd", "build.py")).read()


# ---------------------------------------------------------------- the product never touches the oracle
def test_product_package_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/:
    nothing in the package, the C sources, the headers or the helper scripts does."""
    offenders = []
    for top in ("thrifty_amd", "include", "scripts"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                    text = open(os.path.join(dirpath, f), encoding="utf-8", errors="replace").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "oracle/" in text:
                        offenders.append(os.path.join(dirpath, f))
    assert offenders == []
    # bench.py: the oracle appears only inside the CPU-baseline functions
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^\s*from oracle import", bench_src, re.M):
        enclosing = re.findall(r"^def (\w+)\(", bench_src[:m.start()], re.M)[-1]
        assert enclosing in ("make_oracle", "card_to_toad_leg"), enclosing   # the cpu_baseline legs
    for fn in ("make_oracle", "card_to_toad_leg"):     # ... and only those legs call them
        users = {re.findall(r"^def (\w+)\(", bench_src[:m.start()], re.M)[-1]
                 for m in re.finditer(r"\b%s\(" % fn, bench_src) if not bench_src[:m.start()].endswith("def ")}
        # (_parity_worker: the c3 leg's cpu_baseline -- the oracle over 2048 blocks of that leg)
        assert users <= {"cpu_baseline", "_oracle_worker", "_parity_worker", "main"}, (fn, users)
    header = open(os.path.join(ROOT, "oracle", "thrifty_np.py")).read()
    assert "TEST INFRASTRUCTURE ONLY" in header and "PINNED" in header


@pytest.mark.parametrize("n,h", [(64, 16), (64, 40), (64, 48), (64, 0), (64, 62)])
def test_rawstream_batches_frame_like_block_reader(n, h):
    """RawStream's (lead-in c64 blocks, then contiguous u8 stream with 2H carry) describes
    exactly block_reader's blocks (reference block_data.py:70-98)."""
    import io
    rng = np.random.default_rng(n * 100 + h)
    step = 2 * (n - h)
    data = rng.integers(0, 256, size=step * 23 + 3, dtype=np.uint8).tobytes()
    ref = list(block_data.block_reader(io.BytesIO(data), n, h))
    rs = block_data.RawStream(io.BytesIO(data), n, h)
    got, kinds = [], []
    while True:
        batch = rs.next_batch(5)
        if batch is None:
            break
        kind, stamps, idx, payload = batch
        kinds.append(kind)
        assert len(stamps) == len(idx)
        if kind == "c64":
            got += [(int(i), np.array(b)) for i, b in zip(idx, payload)]
        else:
            a = np.frombuffer(payload, dtype=np.uint8)
            assert a.size == 2 * h + len(idx) * step
            got += [(int(i), block_data.raw_to_complex(a[j * step: j * step + 2 * n]))
                    for j, i in enumerate(idx)]
            del a
    assert len(got) == len(ref)
    assert kinds == sorted(kinds)                      # all lead-in batches come first
    n_lead = min(-(-h // (n - h)), len(ref))           # blocks that still see the zero history
    assert kinds.count("c64") == -(-n_lead // 5)
    for (i, blk), (_, ri, rb) in zip(got, ref):
        assert i == ri
        assert np.array_equal(blk.astype(np.complex128), np.asarray(rb).astype(np.complex128))
    # plain iteration is block_reader
    again = list(block_data.RawStream(io.BytesIO(data), n, h))
    assert len(again) == len(ref) and all(np.array_equal(a[2], b[2]) for a, b in zip(again, ref))


def test_batch_readers_map_regular_files(tmp_path):
    """A regular file is framed straight out of an mmap (no read buffer); the batches are the
    same as through the buffered path a pipe takes."""
    import io
    import mmap
    rng = np.random.default_rng(11)
    for n, h in [(64, 16), (64, 40), (64, 0), (64, 62)]:
        data = rng.integers(0, 256, size=2 * (n - h) * 23 + 5, dtype=np.uint8).tobytes()
        path = tmp_path / ("r_%d_%d.bin" % (n, h))
        path.write_bytes(data)

        def collect(src):
            rs, out = block_data.RawStream(src, n, h), []
            while True:
                batch = rs.next_batch(5)
                if batch is None:
                    return out, rs
                kind, _, idx, payload = batch
                if kind == "c64":
                    out += [(int(i), np.array(x)) for i, x in zip(idx, payload)]
                else:
                    a = np.frombuffer(payload, dtype=np.uint8)
                    step = 2 * (n - h)
                    out += [(int(i), block_data.raw_to_complex(a[j * step: j * step + 2 * n]))
                            for j, i in enumerate(idx)]
        buffered, rb = collect(io.BytesIO(data))
        with open(path, "rb") as f:
            mapped, rm = collect(f)
        assert rb._map is None and isinstance(rm._map, mmap.mmap)
        assert len(buffered) == len(mapped)
        assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(buffered, mapped))
    n = 256
    blocks = [rng.integers(0, 256, size=2 * n, dtype=np.uint8) for _ in range(9)]
    text = "# hdr\nUsing Volk machine: x\n\n" + "".join(
        block_data.card_line(10.0 + i, i * 3, blocks[i]) for i in range(9))
    text = text.replace("\n", "\r\n", 3).encode()
    (tmp_path / "c.card").write_bytes(text)

    def card_collect(src):
        cs, out = block_data.CardStream(src, n), []
        while True:
            batch = cs.next_batch(4)
            if batch is None:
                return out, cs
            stamps, idx, buf, offs = batch
            out += [(s, int(i), bytes(buf[o:o + cs.payload_chars])) for s, i, o in zip(stamps, idx, offs)]
    a, _ = card_collect(io.BytesIO(text))
    with open(tmp_path / "c.card", "rb") as f:
        b, cm = card_collect(f)
        assert isinstance(cm._buf, mmap.mmap)
    assert a == b and len(a) == 9


def test_toad_lines_equal_per_record_serialize():
    """The column-at-a-time .toad formatter writes exactly what building each DetectionResult
    (as Detector._results does) and calling serialize() writes (reference toads_data.py:47-61)."""
    rng = np.random.default_rng(11)
    n = 400
    recs = np.zeros(n, dtype=_native.RECORD_DTYPE)
    recs["block_idx"] = rng.integers(0, 1 << 40, n)
    recs["flags"] = 3
    recs["carrier_bin"] = rng.integers(0, 16384, n)
    recs["corr_sample"] = rng.integers(0, 16384, n)
    scale = 10.0 ** rng.integers(-8, 9, n)
    recs["corr_offset"] = rng.uniform(-0.6, 0.6, n)
    recs["carrier_offset"] = rng.normal(0, 2, n)
    for f in ("corr_energy", "corr_noise", "carrier_energy", "carrier_noise"):
        recs[f] = (rng.uniform(0.1, 10, n) * scale).astype(np.float32)
    recs["corr_offset"][:3] = [0.0, -0.0, 1e-5]
    recs["carrier_energy"][:4] = [1e-4, 9.9e-5, 123456792.0, 1e16]
    stamps = rng.uniform(1.5e9, 1.6e9, n)
    new_len = 12288
    for rxid in (None, 4):
        for otype in (float, np.float32):
            want = []
            for i in range(n):
                r = recs[i]
                car = toads_data.CarrierSyncInfo(int(r["carrier_bin"]), otype(r["carrier_offset"]),
                                                 np.float32(r["carrier_energy"]), np.float32(r["carrier_noise"]))
                cor = toads_data.CorrDetectionInfo(int(r["corr_sample"]), float(r["corr_offset"]),
                                                   float(r["corr_energy"]), float(r["corr_noise"]))
                bi = int(r["block_idx"])
                want.append(toads_data.DetectionResult(stamps[i], bi, new_len * bi + cor.sample + cor.offset,
                                                       car, cor, rxid).serialize())
            got = toads_data.toad_lines(recs, stamps, new_len, rxid=rxid, carrier_offset_type=otype)
            assert got == want
    assert toads_data.toad_lines(recs[:0], stamps[:0], new_len) == []


def test_native_card_framing_equals_python_framing():
    """thr_frame_card (C, host-only) and CardStream._next_batch_py frame the same records: comments,
    banners, blank lines, CRLF, a last line without newline, refills in the middle of a line."""
    import io
    _lib_or_skip()
    rng = np.random.default_rng(3)
    n = 64
    lines = []
    for i in range(57):
        raw = rng.integers(0, 256, 2 * n, dtype=np.uint8)
        ln = block_data.card_line(1.5e9 + 0.001 * i + rng.random() * 1e-3, 1000 + 7 * i, raw)
        if i % 9 == 4:
            ln = ln[:-1] + "\r\n"
        lines.append(ln)
        if i % 11 == 2:
            lines.append("# comment %d\n" % i)
        if i % 13 == 5:
            lines.append("\n")
        if i == 0:
            lines.append("Using Volk machine: avx2_64_mmx\n")
            lines.append("linux; GNU C++ version 4.9\n")
    for tail in ("\n", ""):
        text = ("".join(lines)[:-1] + tail).encode()
        for chunk in (1, 700, 1 << 20):
            for batch in (1, 5, 1000):
                def run(py):
                    cs = block_data.CardStream(io.BytesIO(text), n, chunk_bytes=chunk)
                    out = []
                    while True:
                        b = cs._next_batch_py(batch) if py else cs.next_batch(batch)
                        if b is None:
                            return out
                        stamps, idxs, buf, offs = b
                        for ts, ix, off in zip(stamps, idxs.tolist(), offs.tolist()):
                            out.append((ts, ix, bytes(buf[off:off + cs.payload_chars])))
                a, b = run(False), run(True)
                assert len(a) == 57 and a == b, (tail, chunk, batch)
    cs = block_data.CardStream(io.BytesIO(b"12.5 3 QUJD\n"), n)
    with pytest.raises(ValueError):
        cs.next_batch(4)
    cs = block_data.CardStream(io.BytesIO(b"hello\n"), n)
    with pytest.raises(ValueError):
        cs.next_batch(4)
    # a header that does not parse (`12x.5 7 <payload>`) behind good lines: both framers hand the
    # records in front of it out first, then raise at the line -- the Python framer's fast path too
    good = [block_data.card_line(100.0 + i, i, rng.integers(0, 256, 2 * n, dtype=np.uint8)) for i in range(3)]
    bad = "12x.5 7 " + good[0].split(" ", 2)[2]
    for py in (False, True):
        cs = block_data.CardStream(io.BytesIO("".join(good + [bad] + good).encode()), n)
        first = cs._next_batch_py(16) if py else cs.next_batch(16)
        assert first is not None and first[1].tolist() == [0, 1, 2], py
        with pytest.raises(ValueError):
            cs._next_batch_py(16) if py else cs.next_batch(16)


def test_bench_gpus_n_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no torchrun around it re-launches itself under
    torch.distributed.run (what the driver's launch line does); on a box without an MI355X every
    rank then refuses to run -- there is no CPU fallback to time."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: the GPU suite runs this for real")
    except ImportError:
        pytest.skip("no torch")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                          "--steps", "1", "--warmup", "0", "--legs", "none", "--cpu-seconds", "0"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0
    assert res.stderr.count("bench.py needs an MI355X; there is no CPU fallback") >= 1
    assert "torch.distributed" in res.stderr          # the ranks were started by the launcher
    assert not [ln for ln in res.stdout.split("\n") if ln.startswith("{")]     # and no line was fabricated


def test_a_flag_change_rebuild_leaves_no_object_of_the_old_flags_behind(tmp_path, monkeypatch):
    """thrifty_amd/build.py: when THR_EXTRA_CFLAGS differ from the stamp, every object and the
    library go BEFORE anything is compiled and the stamp is written LAST, after the link -- an
    interrupted rebuild is still seen as 'flags differ' by the next one, never linked from leftovers."""
    import subprocess
    from thrifty_amd import build
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for name in build.SOURCES + [h for h in build.HEADERS if not h.startswith("..")]:
        (csrc / name).write_text("// stub\n")
    inc = tmp_path / "include"
    inc.mkdir()
    (inc / "thrifty_hip.h").write_text("// stub\n")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    monkeypatch.setattr(build, "HEADERS", [h if not h.startswith("..") else str(inc / "thrifty_hip.h")
                                           for h in build.HEADERS])
    lib = tmp_path / "libthriftyhip.so"
    monkeypatch.setattr(build, "LIB", str(lib))
    for src in build.SOURCES:                                     # objects + library of the "old" flags
        (csrc / src.replace(".hip", ".o")).write_text("old object")
    lib.write_text("old library")
    (csrc / ".build_flags").write_text("-DOLD")
    seen = {}

    class Boom(Exception):
        pass

    def fake_popen(cmd):
        # the first compile of the rebuild: by now nothing of the old build may be left, and the
        # stamp must not yet name the new flags
        seen["objects"] = sorted(p.name for p in csrc.glob("*.o"))
        seen["lib"] = lib.exists()
        seen["stamp"] = (csrc / ".build_flags").exists()
        raise Boom()                                              # ... and the rebuild is interrupted here

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setenv("THR_EXTRA_CFLAGS", "-DNEW")
    with pytest.raises(Boom):
        build.build_native()
    assert seen == {"objects": [], "lib": False, "stamp": False}
    assert build.needs_build()                                    # the next build starts from scratch again
