"""GPU tests of the drop-in operator API (thrifty_amd.detect) -- they read like the
reference's usage: build DetectorSettings, iterate a Detector over card_reader
blocks, print result.serialize()."""
import io

import numpy as np
import pytest

from thrifty_amd import block_data
from thrifty_amd.detect import (Detector, DetectorSettings, MultiTemplateDetector,
                                SummaryLineFormatter, detector_cli)

pytestmark = pytest.mark.gpu


def settings_of(g, template=None):
    tpl = g["template"] if template is None else template
    return DetectorSettings(int(g["block_len"]), int(g["history_len"]), int(np.asarray(tpl).shape[-1]),
                            tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                            tpl, tuple(g["corr_thresh"]))


def card_text(g):
    return "# synthetic\n" + "".join(
        block_data.card_line(1000.0 + i, int(g["block_idx"][i]), g["blocks"][i])
        for i in range(len(g["blocks"])))


def assert_toad_close(got_lines, want_text):
    want_lines = str(want_text).split("\n") if len(str(want_text)) else []
    assert len(got_lines) == len(want_lines)
    for got, want in zip(got_lines, want_lines):
        a, b = got.split(), want.split()
        assert a[0] == b[0] and a[2] == b[2]                      # rxid, block
        assert float(a[1]) == float(b[1])                          # timestamp
        assert a[4] == b[4] and a[8] == b[8]                       # corr sample, carrier bin: exact
        np.testing.assert_allclose(float(a[3]), float(b[3]), atol=2e-4, rtol=0)      # soa
        np.testing.assert_allclose(float(a[5]), float(b[5]), atol=1e-4, rtol=1e-4)   # offset
        for k in (6, 7, 10, 11):                                   # energies / noises
            np.testing.assert_allclose(float(a[k]), float(b[k]), rtol=1e-4)
        np.testing.assert_allclose(float(a[9]), float(b[9]), atol=1e-3, rtol=0)      # carrier offset


@pytest.mark.parametrize("name,batch", [("c2", 5), ("c2", 256), ("c1", 7)])
def test_iterating_detector_reproduces_reference_toad(golden, name, batch):
    g = golden(name)
    det = Detector(settings_of(g), block_data.card_reader(io.StringIO(card_text(g))),
                   rxid=int(g["rxid"]), batch_size=batch)
    lines, n = [], 0
    for i, (detected, result) in enumerate(det):
        n += 1
        assert result.block == g["block_idx"][i]                   # input order preserved
        assert detected == bool(g["det"][i])
        assert (result.corr_info is not None) == bool(g["carrier_det"][i])
        if result.corr_info is None:
            assert result.soa is None
        if detected:
            lines.append(result.serialize())
    assert n == len(g["blocks"])
    assert_toad_close(lines, g["toad"])


@pytest.mark.parametrize("name", ["c2", "c1", "small"])
def test_batch_built_results_serialize_like_the_library_and_like_python(golden, name):
    """The results of the iteration are built a batch at a time in C (thrifty_amd._fastresults) and
    serialize() of an untouched detected one is thr_format_toad's line: it must be the text the library
    loop writes (iter_toad_lines) AND the text the reference's Python formatting gives for the same
    attribute values (a result re-built through the reference's constructor formats in Python)."""
    from thrifty_amd import toads_data
    g = golden(name)
    st = settings_of(g)
    items = [(1000.0 + 0.25 * i, int(g["block_idx"][i]), g["blocks"][i]) for i in range(len(g["blocks"]))]
    with Detector(st, iter(items), rxid=int(g["rxid"]), batch_size=9) as det:
        got = list(det)
    with Detector(st, iter(items), rxid=int(g["rxid"]), batch_size=9) as det:
        want = [ln for batch in det.iter_toad_lines() for ln in batch]
    fast = [res.serialize() for detected, res in got if detected]
    assert fast == want and len(fast) == int(np.sum(g["det"]))
    for detected, res in got:
        assert isinstance(res, toads_data.DetectionResult)
        if not detected:
            continue
        assert res._serialize_fast() is not None
        plain = toads_data.DetectionResult(res.timestamp, res.block, res.soa, res.carrier_info, res.corr_info,
                                           res.rxid, res.txid)
        assert plain._serialize_fast() is None and plain.serialize() == res.serialize()
        assert type(res.carrier_info.energy) is np.float32 and type(res.corr_info.energy) is float
        assert res.soa == det.new_len * res.block + res.corr_info.sample + res.corr_info.offset


def test_single_block_detect_and_yield_data(golden):
    g = golden("c2")
    st = settings_of(g)
    det = Detector(st, None, rxid=0, yield_data=True)
    assert det.new_len == 16384 - 4096
    assert det.soa_estimate.window == (1537, 13825)
    blk = block_data.IQBlock(block_data.raw_to_complex(g["blocks"][0]))   # complex Signal, no raw
    detected, res, xhat, corr = det.detect(1000.0, int(g["block_idx"][0]), blk)
    assert detected and res.corr_info.sample == g["sample"][0]
    assert xhat.shape == (16384,) and corr.shape == (16384 - 1023 + 1,)
    assert int(np.argmax(np.abs(corr[1537:13825]))) + 1537 == g["sample"][0]
    np.testing.assert_allclose(np.mean(np.abs(xhat) ** 2), g["xhat_energy"][0], rtol=1e-5)
    # noise-only block: 4-tuple with Nones (detect.py:70-78)
    k = int(np.flatnonzero(~g["carrier_det"])[0])
    detected, res, xhat, corr = det.detect(1.0, 0, block_data.IQBlock(
        block_data.raw_to_complex(g["blocks"][k]), g["blocks"][k]))
    assert not detected and res.corr_info is None and xhat is None and corr is None
    line = SummaryLineFormatter(2.4e6, 16384)(detected, res)
    assert line.startswith("blk=0; carrier: no ")


def test_sync_and_soa_estimate_are_callable_like_the_references(golden):
    """The body of the reference's Detector.detect (detect.py:60-78), written against the two
    sub-objects as an analysis script would: `shifted_fft, carrier_info = det.sync(block)`, then
    `detected, corr_info, corr = det.soa_estimate(shifted_fft)` -- same values as det.detect(),
    as the goldens and as the oracle's intermediates."""
    from oracle import thrifty_np as onp
    g = golden("c2")
    st = settings_of(g)
    det = Detector(st, None, rxid=0)
    assert det.sync.thresh_coeffs == st.carrier_thresh and det.sync.window == st.carrier_window
    assert det.sync.weights is None and det.soa_estimate.corr_len == 16384 - 1023 + 1
    assert det.soa_estimate.thresh_coeffs == st.corr_thresh
    orc = onp.OracleDetector(16384, 4096, g["template"], tuple(g["carrier_thresh"]),
                             tuple(int(v) for v in g["carrier_window"]), tuple(g["corr_thresh"]))
    seen = 0
    for i in range(len(g["blocks"])):
        if g["index_error"][i]:
            continue
        blk = block_data.IQBlock(block_data.raw_to_complex(g["blocks"][i]), g["blocks"][i])
        shifted_fft, cinfo = det.sync(blk)
        assert cinfo.bin == g["cbin"][i]
        np.testing.assert_allclose(cinfo.energy, g["cenergy"][i], rtol=2e-5)
        np.testing.assert_allclose(cinfo.noise, g["cnoise"][i], rtol=2e-5)
        if not g["carrier_det"][i]:
            assert shifted_fft is None and cinfo.offset == 0
            with pytest.raises(NotImplementedError):
                det.soa_estimate(np.zeros(16384, dtype=np.complex64))
            continue
        np.testing.assert_allclose(cinfo.offset, g["coff"][i], atol=2e-4)
        detected, sinfo, corr = det.soa_estimate(shifted_fft)
        assert detected == bool(g["det"][i]) and sinfo.sample == g["sample"][i]
        np.testing.assert_allclose(sinfo.energy, g["energy"][i], rtol=2e-5)
        np.testing.assert_allclose(sinfo.offset, g["soff"][i] if detected else 0, atol=5e-6)
        assert corr.shape == (16384 - 1023 + 1,)
        assert int(np.argmax(np.abs(corr[1537:13825]))) + 1537 == sinfo.sample
        (res,), ((xh, co),) = orc.detect_u8(0, g["blocks"][i], want_data=True)
        # (the shift follows each side's own carrier-offset estimate, equal to <= 2e-4 bins: a phase
        # ramp of up to 2 pi 2e-4 / sqrt(12) over the block on top of the transform's 1e-6)
        assert np.linalg.norm(shifted_fft - xh) / np.linalg.norm(xh) < 5e-5
        assert np.linalg.norm(corr - co) / np.linalg.norm(co) < 5e-5
        d2, r2 = det.detect(0.0, 0, blk)                     # the one-call form agrees
        assert d2 == detected and r2.corr_info.sample == sinfo.sample and r2.corr_info.energy == sinfo.energy
        with pytest.raises(NotImplementedError):             # only the latest sync()'s spectrum
            det.soa_estimate(shifted_fft.copy())
        seen += 1
    assert seen >= 4
    with pytest.raises(NotImplementedError):
        det.sync.detector(np.ones(16384, dtype=np.float32))


def test_index_error_is_mirrored(golden):
    g = golden("c2_straddle")
    det = Detector(settings_of(g), None)
    bad = int(np.flatnonzero(g["index_error"])[0])
    with pytest.raises(IndexError):
        det.detect(0.0, 0, g["blocks"][bad])


def test_detector_cli_end_to_end(golden, tmp_path, capsys):
    g = golden("c2")
    np.save(tmp_path / "template.npy", g["template"])
    (tmp_path / "detector.cfg").write_text(
        "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
        "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
        "template: %s\n" % (tmp_path / "template.npy"))
    (tmp_path / "rx.card").write_text(card_text(g))
    detector_cli(Detector, argv=[str(tmp_path / "rx.card"), "-o", str(tmp_path / "rx.toad"),
                                 "-c", str(tmp_path / "detector.cfg")])
    assert_toad_close((tmp_path / "rx.toad").read_text().strip().split("\n"), g["toad"])
    summary = capsys.readouterr().out.strip().split("\n")
    assert len(summary) == len(g["blocks"]) and summary[0].startswith("blk=5; carrier: yes")


def test_quiet_cli_and_detections_only_iteration(golden, tmp_path, capsys):
    """--quiet skips the per-block objects of undetected blocks: same .toad, no summary."""
    g = golden("c2")
    np.save(tmp_path / "template.npy", g["template"])
    (tmp_path / "detector.cfg").write_text(
        "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
        "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
        "template: %s\n" % (tmp_path / "template.npy"))
    (tmp_path / "rx.card").write_text(card_text(g))
    detector_cli(Detector, argv=[str(tmp_path / "rx.card"), "-o", str(tmp_path / "rx.toad"), "--quiet",
                                 "-c", str(tmp_path / "detector.cfg")])
    assert_toad_close((tmp_path / "rx.toad").read_text().strip().split("\n"), g["toad"])
    assert capsys.readouterr().out.strip() == ""
    # library use: batches without any detection contribute nothing and do not end the iteration
    st = DetectorSettings(16384, 4096, len(g["template"]), (0, 15, 0), (7, 110), g["template"], (0, 15, 0))
    det = Detector(st, block_data.CardStream(io.BytesIO(card_text(g).encode()), 16384), rxid=0, batch_size=2)
    det.only_detections = True
    got = list(det)
    assert all(d for d, _ in got) and len(got) == int(g["det"].sum())
    assert [r.block for _, r in got] == [int(b) for b, d in zip(g["block_idx"], g["det"]) if d]


def test_multi_template_detector(golden):
    gs = [golden("c5_tx%d" % i) for i in range(4)]
    tpls = np.stack([g["template"] for g in gs])
    st = settings_of(gs[0], tpls)
    items = [(1000.0 + i, int(gs[0]["block_idx"][i]), gs[0]["blocks"][i]) for i in range(12)]
    out = list(MultiTemplateDetector(st, iter(items), rxid=0, batch_size=5))
    assert len(out) == 12 and all(len(o) == 4 for o in out)
    for t, g in enumerate(gs):
        lines = [o[t][1].serialize() for o in out if o[t][0]]
        for ln in lines:
            assert ln.split()[1] == str(t)                          # txid column
        assert_toad_close([" ".join(ln.split()[:1] + ln.split()[2:]) for ln in lines], g["toad"])


def test_raw_stream_mode_matches_oracle(golden, tmp_path):
    """`thrifty detect --raw`: overlapping blocks cut from a raw u8 stream (block_reader,
    reference block_data.py:70-98), first history = 0.0, through the CLI."""
    from oracle import thrifty_np as onp
    from thrifty_amd import synth
    g = golden("c2")
    n, h = 16384, 4096
    new = n - h
    tpl = g["template"]
    rng = np.random.default_rng(31)
    # a continuous stream: noise with three bursts at known stream positions
    nblk = 6
    stream = (rng.normal(0, 0.02, new * nblk) + 1j * rng.normal(0, 0.02, new * nblk))
    ook = 0.3 * (np.asarray(tpl, float) + 1) / 2
    for start, car in ((9000, 33.3), (30000, 71.8), (52000, 55.1)):
        k = np.arange(len(tpl))
        stream[start:start + len(tpl)] += ook * np.exp(2j * np.pi * car * (k + start) / n)
    raw = synth.quantise_iq(stream)
    np.save(tmp_path / "template.npy", tpl)
    (tmp_path / "detector.cfg").write_text(
        "rxid: 9\nsample_rate: 2.4M\nblock_size: %d\nblock_history: %d\ncarrier_window: 7 - 110\n"
        "carrier_threshold: 15 * snr\ncorr_threshold: 15*snr\ntemplate: %s\n" % (n, h, tmp_path / "template.npy"))
    (tmp_path / "rx.raw").write_bytes(raw.tobytes())
    detector_cli(Detector, argv=[str(tmp_path / "rx.raw"), "--raw", "--quiet", "-o", str(tmp_path / "rx.toad"),
                                 "-c", str(tmp_path / "detector.cfg")])
    got = [ln.split() for ln in (tmp_path / "rx.toad").read_text().strip().split("\n")]
    # oracle over the same framing
    orc = onp.OracleDetector(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0))
    want = []
    for ts, idx, blk in block_data.block_reader(io.BytesIO(raw.tobytes()), n, h):
        (res,) = orc.detect_block(idx, np.asarray(blk))
        if res.detected:
            want.append((idx, res))
    assert len(got) == len(want) == 3
    for a, (idx, res) in zip(got, want):
        assert int(a[0]) == 9 and int(a[2]) == idx and int(a[4]) == res.corr.sample
        np.testing.assert_allclose(float(a[3]), res.soa, atol=2e-4)
        np.testing.assert_allclose(float(a[6]), res.corr.energy, rtol=1e-4)
    # each burst is reported exactly once although blocks overlap by `history` samples
    soas = sorted(float(a[3]) for a in got)
    np.testing.assert_allclose(soas, [9000 + h, 30000 + h, 52000 + h], atol=1.0)
