"""Pin the CPU oracle (oracle/thrifty_np.py) against the reference.

(1) fixtures produced by running the reference itself (tests/golden/*.npz);
(2) the known-answer tables of the reference's own unit tests, re-expressed
    against the oracle (reference tests/test_carrier_detect.py:11-72,
    test_carrier_sync.py:12-65, test_soa_estimator.py:13-109,
    test_block_data.py:13-37, test_util.py:11-16).
"""
import os

import numpy as np
import pytest
import scipy.signal

from oracle import thrifty_np as onp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FIXTURES = ["c2", "c2_negwin", "c2_straddle", "c2_stddev", "c2_fullwin",
            "c5_tx0", "c5_tx3", "c1", "c3", "small"]


def make_oracle(g):
    return onp.OracleDetector(int(g["block_len"]), int(g["history_len"]), g["template"],
                              tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                              tuple(g["corr_thresh"]))


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_matches_reference_records(golden, name):
    g = golden(name)
    orc = make_oracle(g)
    lines = []
    for i, raw in enumerate(g["blocks"]):
        if g["index_error"][i]:
            with pytest.raises(IndexError):
                orc.detect_u8(int(g["block_idx"][i]), raw)
            continue
        (res,), ((xhat, corr),) = orc.detect_u8(int(g["block_idx"][i]), raw, want_data=True)
        car = res.carrier
        assert car.bin == g["cbin"][i]
        assert car.detected == bool(g["carrier_det"][i])
        assert res.detected == bool(g["det"][i])
        np.testing.assert_allclose(car.offset, g["coff"][i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(car.energy, g["cenergy"][i], rtol=1e-7)
        np.testing.assert_allclose(car.noise, g["cnoise"][i], rtol=1e-7)
        if car.detected:
            cs = res.corr
            assert cs.sample == g["sample"][i]
            np.testing.assert_allclose(cs.offset, g["soff"][i], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(cs.energy, g["energy"][i], rtol=1e-10)
            np.testing.assert_allclose(cs.noise, g["noise"][i], rtol=1e-10)
            np.testing.assert_allclose(res.soa, g["soa"][i], rtol=0, atol=1e-7)
            np.testing.assert_allclose(np.mean(np.abs(xhat) ** 2), g["xhat_energy"][i], rtol=1e-10)
            if 0 < cs.sample < len(corr) - 1:
                np.testing.assert_allclose(np.abs(corr)[cs.sample - 1:cs.sample + 2],
                                           g["corr3"][i], rtol=1e-9)
        if res.detected:
            lines.append(onp.toad_line(int(g["rxid"]), 1000.0 + i, int(g["block_idx"][i]), res))
    assert "\n".join(lines) == str(g["toad"])


# ---- reference tests/test_carrier_detect.py:11-22 -------------------------
@pytest.mark.parametrize("start,stop,length,expected", [
    (50, 100, 1024, (50, 100)), (0, -1, 1024, (0, 1023)),
    (-10, 10, 1024, (1014, 1034)), (-1, 0, 1024, (1023, 1024))])
def test_window_to_indices(start, stop, length, expected):
    assert onp.window_to_indices(start, stop, length) == expected


# ---- reference tests/test_carrier_detect.py:25-72 -------------------------
@pytest.mark.parametrize("fmin,fmax,freq,expected", [
    (-81e3, -79e3, -80e3, True), (-81e3, -79e3, -79.1e3, True), (-81e3, -79e3, -80.9e3, True),
    (-81e3, -79e3, -82e3, False), (-81e3, -79e3, -78e3, False), (-81e3, -79e3, 0.0, False),
    (79e3, 81e3, 80e3, True), (79e3, 81e3, 79.1e3, True), (79e3, 81e3, 80.9e3, True),
    (79e3, 81e3, 82e3, False), (79e3, 81e3, 78e3, False), (79e3, 81e3, -80e3, False),
    (79e3, 81e3, 0.0, False), (-10e3, 5e3, 0.0, True), (-10e3, 5e3, -9.9e3, True),
    (-10e3, 5e3, 4.9e3, True), (-10e3, 5e3, 6e3, False), (-10e3, 5e3, -11e3, False)])
def test_carrier_window_semantics(fmin, fmax, freq, expected):
    n, w, fs = 8192, 2085, 2.2e6
    binf = fs / n
    tone = np.exp(2j * np.pi * freq * np.arange(w) / fs)
    mag = np.abs(np.fft.fft(np.concatenate([tone, np.zeros(n - w)])))
    det = onp.carrier_detect(mag, (500.0 ** 2, 0.0, 0.0), (int(fmin / binf), int(fmax / binf)))[0]
    assert det == expected


# ---- reference tests/test_carrier_sync.py:12-65 ----------------------------
@pytest.mark.parametrize("size,freq,shift", [(128, 0, 0), (128, -32, 32), (128, 32, 16),
                                             (128, -10.5, 0.5), (128, 8.3, -8.3)])
def test_shift_and_fft(size, freq, shift):
    sig = np.exp(2j * np.pi * np.arange(size) / size * freq)
    want = np.fft.fft(np.exp(2j * np.pi * np.arange(size) / size * (freq + shift)))
    np.testing.assert_allclose(np.abs(onp.shift_and_fft(sig, shift)), np.abs(want), atol=1e-6, rtol=1e-6)


def test_dirichlet_known_answer():
    want = np.array([-0.1711, 0.0164, 0.3164, 0.6468, 0.9034, 1., 0.9034, 0.6468, 0.3164, 0.0164, -0.1711])
    np.testing.assert_allclose(onp.dirichlet(np.arange(-5, 6), 8192, 2015), want, rtol=2e-3)


@pytest.mark.parametrize("offset", [-0.51, -0.5, -0.25, -0.1263, -0.1, 0., 0.001, 0.2, 0.4995, 0.56])
def test_dirichlet_fit_recovers_offset(offset):
    peak, n, w = 10, 8192, 2024
    f = (offset + peak) * w / n
    tone = np.exp(2j * np.pi * np.arange(w) / w * f)
    mag = np.abs(np.fft.fft(np.concatenate([tone, np.zeros(n - w)])))
    np.testing.assert_allclose(onp.dirichlet_fit(mag, peak, n, w)[1], offset, atol=1e-8, rtol=1e-8)


# ---- reference tests/test_soa_estimator.py:13-109 --------------------------
TPL31 = np.array([1, 1, 1, 1, 1, -1, -1, -1, 1, 1, -1, 1, 1, 1, -1, 1,
                  -1, 1, -1, -1, -1, -1, 1, -1, -1, 1, -1, 1, 1, -1, -1])


def _ook_block(pos, n=64):
    b = np.zeros(n)
    sig = (TPL31 + 1) / 2
    end = min(n, pos + len(sig))
    b[pos:end] += sig[:end - pos]
    return b


@pytest.mark.parametrize("pos", [0, 1, 10, 33, 34, 63])
def test_despread_peaks_and_crosscheck(pos):
    bank = onp.TemplateBank(TPL31, 64, len(TPL31))
    blk = _ook_block(pos)
    corr = onp.despread(np.fft.fft(blk), bank)
    assert len(corr) == 64 - 31 + 1
    mag = np.abs(corr)
    if pos <= 33:
        pk = int(np.argmax(mag))
        assert pk == pos and mag[pk] >= 15.9
        assert np.all(np.delete(mag, pk) < 5.1)
    else:
        assert np.all(mag < 5.1)
    np.testing.assert_allclose(corr, scipy.signal.correlate(blk, TPL31, mode="valid"), atol=1e-12, rtol=1e-12)


@pytest.mark.parametrize("params,expected", [((64, 31, 32), (0, 33)), ((64, 32, 32), (0, 32)),
                                             ((64, 33, 32), (1, 32)), ((64, 63, 32), (16, 17))])
def test_unique_window(params, expected):
    assert onp.unique_window(*params) == expected


@pytest.mark.parametrize("idx,n,window,expected", [(0, 33, (0, 33), True), (32, 33, (0, 33), True),
                                                   (1, 33, (1, 32), True), (31, 33, (1, 32), True),
                                                   (0, 33, (1, 32), False), (32, 33, (1, 32), False)])
def test_corr_peak_window(idx, n, window, expected):
    mag = np.zeros(n)
    mag[idx] = 100
    pk, val = onp.corr_peak(mag, window)
    assert (val > 99) == expected
    if expected:
        assert pk == idx and val == 100


# ---- reference tests/test_block_data.py:13-37, test_util.py:11-16 ----------
def test_iq_conversion_known_answers():
    raw = np.array([0, 0, 127, 128, 255, 255], dtype=np.uint8)
    cplx = np.array([-0.9953 - 0.9953j, -0.0031 + 0.0047j, 0.9969 + 0.9969j], dtype=np.complex64)
    np.testing.assert_allclose(onp.iq_u8_to_c64(raw), cplx, rtol=1e-2)
    np.testing.assert_array_equal(onp.c64_to_iq_u8(cplx), raw)
    every = np.arange(256, dtype=np.uint8)
    np.testing.assert_array_equal(onp.c64_to_iq_u8(onp.iq_u8_to_c64(every)), every)


@pytest.mark.parametrize("num", [15, 16])
def test_fft_bin(num):
    got = np.array([onp.fft_bin(i, num) for i in range(num)])
    np.testing.assert_array_equal(got, np.fft.fftfreq(num, 1. / num))


# ---- PreshiftDetector variant: fixtures from the reference's
# ---- experimental/detect_preshift.py (tests/golden/make_golden_preshift.py)
PRESHIFT_FIXTURES = ["preshift_c2", "preshift_c2_straddle", "preshift_c2_stddev", "preshift_c1",
                     "preshift_small",
                     # the reference's other three-point carrier interpolators
                     "preshift_c2_none", "preshift_c2_gaussian", "preshift_c2_cosine",
                     "preshift_c2_straddle_gaussian"]


@pytest.mark.parametrize("name", PRESHIFT_FIXTURES)
def test_preshift_oracle_matches_reference_records(golden, name):
    g = golden(name)
    orc = onp.OraclePreshiftDetector(
        int(g["block_len"]), int(g["history_len"]), g["template"], tuple(g["carrier_thresh"]),
        tuple(int(v) for v in g["carrier_window"]), tuple(g["corr_thresh"]), num=int(g["num"]),
        interpolator=str(g["interpolator"]) if "interpolator" in g.files else "parabolic")
    from conftest import golden_blocks
    lines = []
    for i, raw in enumerate(golden_blocks(g)):
        if g["index_error"][i]:
            with pytest.raises(IndexError):
                orc.detect_u8(int(g["block_idx"][i]), raw)
            continue
        res = orc.detect_u8(int(g["block_idx"][i]), raw)
        car = res.carrier
        assert car.bin == g["cbin"][i] and car.detected == bool(g["carrier_det"][i])
        assert res.detected == bool(g["det"][i])
        assert car.energy == g["cenergy"][i] and car.noise == g["cnoise"][i]
        if not car.detected:
            continue
        if "interpolator" in g.files and str(g["interpolator"]) == "none":
            assert car.offset == 0 and isinstance(car.offset, int)
        else:
            assert isinstance(car.offset, np.float32) and car.offset == g["coff"][i]
        assert orc.last[1] == g["frac_shift"][i]
        cs = res.corr
        assert cs.sample == g["sample"][i]
        assert cs.offset == g["soff"][i] and cs.energy == g["energy"][i] and cs.noise == g["noise"][i]
        if res.detected:
            lines.append(onp.toad_line(int(g["rxid"]), 1000.0 + i, int(g["block_idx"][i]), res))
    assert "\n".join(lines) == str(g["toad"])          # byte for byte what the reference prints


# ---- identify: fixtures from the reference's identify.py
# ---- (tests/golden/make_golden_identify.py)
def freqmap_of(g):
    fm = {}
    for rx, tx, lo, hi in zip(g["map_rx"], g["map_tx"], g["map_lo"], g["map_hi"]):
        fm.setdefault(int(rx), {})[int(tx)] = (float(lo), float(hi))
    return fm


@pytest.mark.parametrize("name", ["identify_auto3", "identify_auto1", "identify_map"])
def test_identify_oracle_matches_reference(golden, name):
    g = golden(name)
    if "edges" in g.files:
        txid, edges = onp.auto_classify(g["rxid"], g["carrier_bin"])
        for j, rx in enumerate(g["edge_rx"]):
            want = g["edges"][g["edge_ptr"][j]:g["edge_ptr"][j + 1]]
            assert np.array_equal(np.asarray(edges[int(rx)], dtype=np.int64), want)
    else:
        txid = onp.classify_by_map(g["rxid"], g["carrier_bin"], g["carrier_offset"], freqmap_of(g))
    assert np.array_equal(txid, g["txid"])
    mask = onp.duplicate_mask(g["rxid"], txid, g["block"], g["timestamp"], g["energy"])
    assert np.array_equal(mask, g["dup_mask"])
    assert np.array_equal(onp.filter_order(mask, g["timestamp"]), g["kept_order"])



INTERPOL = {"none": lambda n, w: onp.no_offset, "parabolic": lambda n, w: onp.parabolic_offset,
            "gaussian": lambda n, w: onp.gaussian_offset, "cosine": lambda n, w: onp.cosine_offset,
            "parabole_fit6": lambda n, w: onp.parabole_fit_offset(6),
            "corr_parabolic4": lambda n, w: onp.corr_parabolic_offset(4, n, w)}


@pytest.mark.parametrize("method", sorted(INTERPOL))
def test_replaced_interpolator_reproduces_the_references_interpolation_detector(method):
    """`sync.interpolator = fn` (reference experimental/detect_carrier_interpol.py:17-40): the oracle
    with the same interpolator reproduces the reference's `.toad` text byte for byte, and the type
    of the offset (none() and cosine's early return hand back the Python int 0)."""
    g = np.load(os.path.join(GOLDEN_DIR, "interpol_c2_%s.npz" % method), allow_pickle=False)
    src = np.load(os.path.join(GOLDEN_DIR, str(g["src"]) + ".npz"), allow_pickle=False)
    n, w = int(src["block_len"]), len(src["template"])
    orc = onp.OracleDetector(n, int(src["history_len"]), src["template"], tuple(src["carrier_thresh"]),
                             tuple(int(v) for v in src["carrier_window"]), tuple(src["corr_thresh"]),
                             interpolator=INTERPOL[method](n, w))
    lines = []
    for i, raw in enumerate(src["blocks"]):
        (res,) = orc.detect_u8(int(src["block_idx"][i]), raw)
        assert res.carrier.bin == g["cbin"][i] and res.carrier.detected == g["carrier_det"][i]
        assert res.detected == g["det"][i]
        if res.carrier.detected:
            assert isinstance(res.carrier.offset, int) == bool(g["coff_is_int"][i])
            assert res.carrier.offset == g["coff"][i] and res.corr.sample == g["sample"][i]
        if res.detected:
            lines.append(onp.toad_line(int(src["rxid"]), 1000.0 + i, int(src["block_idx"][i]), res))
    assert "\n".join(lines) == str(g["toad"])


XCORR = {"none": lambda t: onp.xcorr_none, "parabolic": lambda t: onp.xcorr_parabolic,
         "gaussian": lambda t: onp.xcorr_gaussian, "cosine": lambda t: onp.xcorr_cosine,
         "autocorr": onp.xcorr_autocorr, "maximise": onp.xcorr_maximise}
XCORR_CASES = [("c2", "none"), ("c2", "parabolic"), ("c2", "cosine"), ("c2", "gaussian"), ("c2", "maximise"),
               ("c1", "autocorr"), ("c1", "maximise"), ("c1", "parabolic")]


@pytest.mark.parametrize("src_name,method", XCORR_CASES)
def test_replaced_correlation_interpolator_reproduces_the_references_experiment(src_name, method):
    """`soa_estimate.interpolate = fn` / IterativeSoaEstimator (reference experimental/
    detect_xcorr_interpol.py:20-62): the oracle with the same interpolator reproduces the reference's
    `.toad` text byte for byte (the three-point ones and `none`) or to the optimiser's tolerance, the
    +-0.6 clip (c1 / parabolic hits it twice) and the int 0 of none()."""
    g = np.load(os.path.join(GOLDEN_DIR, "xcorr_%s_%s.npz" % (src_name, method)), allow_pickle=False)
    src = np.load(os.path.join(GOLDEN_DIR, src_name + ".npz"), allow_pickle=False)
    assert str(g["src"]) == src_name and str(g["method"]) == method
    orc = onp.OracleDetector(int(src["block_len"]), int(src["history_len"]), src["template"],
                             tuple(src["carrier_thresh"]), tuple(int(v) for v in src["carrier_window"]),
                             tuple(src["corr_thresh"]), interpolate=XCORR[method](src["template"]))
    exact = method not in ("autocorr", "maximise")
    lines = []
    for i, raw in enumerate(src["blocks"]):
        (res,) = orc.detect_u8(int(src["block_idx"][i]), raw)
        assert res.carrier.bin == g["cbin"][i] and res.carrier.detected == g["carrier_det"][i]
        assert res.detected == g["det"][i]
        if not res.carrier.detected:
            continue
        assert res.carrier.offset == g["coff"][i] and res.corr.sample == g["sample"][i]
        assert isinstance(res.corr.offset, int) == bool(g["soff_is_int"][i])
        assert abs(res.corr.offset) <= 0.6
        if exact:
            assert res.corr.offset == g["soff"][i]
        else:
            np.testing.assert_allclose(res.corr.offset, g["soff"][i], atol=1e-6)
        if res.detected:
            lines.append(onp.toad_line(int(src["rxid"]), 1000.0 + i, int(src["block_idx"][i]), res))
    if exact:
        assert "\n".join(lines) == str(g["toad"])
    if (src_name, method) == ("c1", "parabolic"):
        assert (np.abs(g["soff"][g["det"]]) == 0.6).sum() == 2


def test_the_fixture_generator_and_the_test_name_the_same_cases():
    assert [list(c) for c in XCORR_CASES] == [list(c) for c in _declared("make_golden_xcorr.py", ["CASES"])["CASES"]]


def _declared(generator, names):
    """Module-level list / dict literals of a fixture generator, read with `ast` -- the generators
    import the reference at import time, which exists in the build container only."""
    import ast
    tree = ast.parse(open(os.path.join(GOLDEN_DIR, generator)).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and getattr(node.targets[0], "id", None) in names:
            out[node.targets[0].id] = ast.literal_eval(node.value)
    assert set(out) == set(names), (generator, sorted(out))
    return out


def test_every_fixture_holds_exactly_the_keys_its_generator_writes():
    """A fixture that predates a change of its generator (a key added, renamed, dropped) would
    still load -- and a test reading the new key would fail only where it runs.  Every committed
    .npz must carry exactly the key set its generator declares (and asserts when it writes)."""
    import glob
    det = _declared("make_golden.py", ["KEYS", "EXTRA_KEYS"])
    pre = _declared("make_golden_preshift.py", ["KEYS", "KEYS_OWN_BLOCKS", "KEYS_SHARED_BLOCKS"])
    ide = _declared("make_golden_identify.py", ["KEYS", "KEYS_AUTO", "KEYS_MAP"])
    itp = _declared("make_golden_interpol.py", ["KEYS"])
    xco = _declared("make_golden_xcorr.py", ["KEYS", "CASES"])
    assert sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN_DIR, "xcorr_*.npz"))) == sorted(
        "xcorr_%s_%s.npz" % tuple(c) for c in xco["CASES"])
    seen = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(path)[:-4]
        have = sorted(np.load(path, allow_pickle=False).files)
        if name.startswith("preshift_"):
            shared = "src" in have
            want = pre["KEYS"] + (pre["KEYS_SHARED_BLOCKS"] if shared else pre["KEYS_OWN_BLOCKS"])
        elif name.startswith("interpol_"):
            want = itp["KEYS"]
        elif name.startswith("xcorr_"):
            want = xco["KEYS"]
        elif name.startswith("identify_"):
            want = ide["KEYS"] + (ide["KEYS_MAP"] if name == "identify_map" else ide["KEYS_AUTO"])
        else:
            want = det["KEYS"] + det["EXTRA_KEYS"].get(name, [])
        assert have == sorted(want), (name, sorted(set(have) ^ set(want)))
        seen += 1
    assert seen >= 30
