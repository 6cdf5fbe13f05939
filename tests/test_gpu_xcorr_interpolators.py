"""A replaced correlation-peak interpolator on the GPU engine: `Detector.soa_estimate.interpolate = fn`,
what the reference's second experiment does (thrifty/experimental/detect_xcorr_interpol.py:20-62 with
the functions of xcorr_interpolators.py:31-112).  The engine keeps its verdicts and peak search; the
callable runs on the host on the correlation magnitudes of the detected blocks (a stage dump per
batch).  Fixtures `xcorr_*` come from running the reference class."""
import io

import numpy as np
import pytest

from thrifty_amd import _native as F
from thrifty_amd.detect import Detector
from oracle import thrifty_np as onp

from test_gpu_detector_api import card_text, settings_of
from test_oracle_golden import XCORR_CASES

pytestmark = pytest.mark.gpu

# the callables assigned here are the ORACLE's restatements: the product ships the hook
# (`soa_estimate.interpolate = fn`), not the reference's interpolators


def host_interpolate(det, src, method):
    """What the reference's class assigns for `method` (detect_xcorr_interpol.py:36-62), from the oracle;
    its callables take (corr_mag, peak_idx, xhat), the hook hands out the first two and keeps the
    block's shifted spectrum in soa_estimate.last_fft."""
    if method == "gaussian":
        return None                              # the engine's own log-parabola: nothing to assign
    fn = {"none": onp.xcorr_none, "parabolic": onp.xcorr_parabolic, "cosine": onp.xcorr_cosine,
          "autocorr": lambda: onp.xcorr_autocorr(src["template"]),
          "maximise": lambda: onp.xcorr_maximise(src["template"])}[method]
    if method in ("autocorr", "maximise"):
        fn = fn()
    return lambda corr_mag, peak: fn(corr_mag, peak, det.soa_estimate.last_fft)


def items_of(src):
    return [(1000.0 + i, int(src["block_idx"][i]), src["blocks"][i]) for i in range(len(src["blocks"]))]


@pytest.mark.parametrize("src_name,method", XCORR_CASES)
def test_interpolation_detector_matches_the_references(golden, src_name, method):
    g, src = golden("xcorr_%s_%s" % (src_name, method)), golden(src_name)
    st = settings_of(src)
    det = Detector(st, iter(items_of(src)), rxid=int(src["rxid"]), batch_size=5)
    fn = host_interpolate(det, src, method)
    if fn is not None:
        det.soa_estimate.interpolate = fn
    assert det._host_soa == (method != "gaussian")          # `gaussian` is the engine's own: the fast path
    got = list(det)
    assert len(got) == len(src["blocks"])
    # float32 magnitudes of another FFT; c1's peak is broad (an oversampled template), its three
    # magnitudes nearly equal: the three-point formulas amplify their rounding
    atol = 2e-3 if src_name == "c1" else 1e-4
    lines = []
    for i, (detected, res) in enumerate(got):
        assert res.carrier_info.bin == g["cbin"][i]
        assert (res.corr_info is not None) == bool(g["carrier_det"][i]) and detected == bool(g["det"][i])
        if res.corr_info is None:
            continue
        np.testing.assert_allclose(res.carrier_info.offset, g["coff"][i], atol=2e-4)
        assert res.corr_info.sample == g["sample"][i]                       # bit-exact SoA sample
        np.testing.assert_allclose(res.corr_info.energy, g["energy"][i], rtol=1e-4)
        np.testing.assert_allclose(res.corr_info.noise, g["noise"][i], rtol=1e-4)
        if not detected:
            assert res.corr_info.offset == 0                                # soa_estimator.py:87
            continue
        assert abs(res.corr_info.offset) <= 0.6
        if method == "none":
            assert isinstance(res.corr_info.offset, int) and res.corr_info.offset == 0
        np.testing.assert_allclose(float(res.corr_info.offset), g["soff"][i], atol=atol)
        np.testing.assert_allclose(res.soa, g["soa"][i], atol=atol + 1e-4)
        assert res.soa == det.new_len * res.block + res.corr_info.sample + res.corr_info.offset
        lines.append(res.serialize())
    want = str(g["toad"]).split("\n")
    assert len(lines) == len(want)
    for a, b in zip(lines, want):
        fa, fb = a.split(), b.split()
        assert fa[:3] == fb[:3] and fa[4] == fb[4] and fa[8] == fb[8]       # rxid ts block | sample | bin
        if method == "none":
            assert fa[5] == fb[5] == "0"                                    # the int 0 prints as "0"
    if (src_name, method) == ("c1", "parabolic"):                           # the +-0.6 clip (soa_estimator.py:16-17, :88)
        assert sum(abs(res.corr_info.offset) == 0.6 for d, res in got if d) == 2


def test_the_engines_own_interpolator_restated_on_the_host_gives_the_engines_records(golden):
    """The log-parabola (soa_estimator.py:159-170) assigned as the interpolator is the default detector computed
    the slow way -- the same offsets to float32 rounding, everything else equal."""
    src = golden("c2")
    st = settings_of(src)
    want = list(Detector(st, iter(items_of(src)), rxid=0))
    slow = Detector(st, iter(items_of(src)), rxid=0, batch_size=7)
    seen = []

    def gaussian(corr_mag, peak):
        seen.append((corr_mag.dtype, corr_mag.shape, peak, slow.soa_estimate.last_fft.shape))
        return onp.xcorr_gaussian(corr_mag, peak)

    slow.soa_estimate.interpolate = gaussian
    assert slow._host_soa and not slow._host_interp and slow.soa_estimate.interpolate is gaussian
    got = list(slow)
    assert len(got) == len(want) and len(seen) == sum(d for d, _ in want)   # called for DETECTED blocks only
    corr_len = 16384 - len(src["template"]) + 1
    assert all(s[0] == np.float32 and s[1] == (corr_len,) and s[3] == (16384,) for s in seen)
    assert [s[2] for s in seen] == [r.corr_info.sample for d, r in want if d]
    assert slow.soa_estimate.last_fft is None
    for (d1, r1), (d2, r2) in zip(want, got):
        assert d1 == d2 and r1.carrier_info == r2.carrier_info
        if r1.corr_info is None:
            assert r2.corr_info is None
            continue
        assert r1.corr_info.sample == r2.corr_info.sample and r1.corr_info.energy == r2.corr_info.energy
        np.testing.assert_allclose(r2.corr_info.offset, r1.corr_info.offset, atol=2e-5)
        if d1:
            np.testing.assert_allclose(r2.soa, r1.soa, atol=1e-4)


def test_both_stages_replaced_follow_the_oracle(golden):
    """`sync.interpolator` AND `soa_estimate.interpolate` replaced: the correlation the second
    callable sees is the one of the block shifted by the first callable's offset
    (thr_debug_stage_offsets)."""
    src = golden("c2")
    st = settings_of(src)
    det = Detector(st, iter(items_of(src)), rxid=0, batch_size=6)
    det.sync.interpolator = onp.parabolic_offset
    det.soa_estimate.interpolate = onp.xcorr_cosine
    assert det._host_interp and det._host_soa
    got = list(det)
    orc = onp.OracleDetector(16384, int(src["history_len"]), src["template"], tuple(src["carrier_thresh"]),
                             tuple(int(v) for v in src["carrier_window"]), tuple(src["corr_thresh"]),
                             interpolator=onp.parabolic_offset, interpolate=onp.xcorr_cosine)
    n_det = 0
    for (detected, res), raw, bi in zip(got, src["blocks"], src["block_idx"]):
        (want,) = orc.detect_u8(int(bi), raw)
        assert detected == want.detected and res.carrier_info.bin == want.carrier.bin
        if not want.carrier.detected:
            continue
        np.testing.assert_allclose(float(res.carrier_info.offset), want.carrier.offset, atol=5e-5)
        assert res.corr_info.sample == want.corr.sample
        np.testing.assert_allclose(res.corr_info.energy, want.corr.energy, rtol=1e-4)
        if detected:
            n_det += 1
            np.testing.assert_allclose(float(res.corr_info.offset), want.corr.offset, atol=1e-4)
            np.testing.assert_allclose(res.soa, want.soa, atol=2e-4)
    assert n_det >= 15
    # the dump with given offsets is the plain dump when the offsets are the fit's own
    eng = F.Engine(16384, 4096, src["template"], (0, 15, 0), (7, 110), (0, 15, 0), max_batch=32)
    rec = eng.detect(src["blocks"], src["block_idx"])[:, 0]
    x0, c0 = eng.debug_stage(src["blocks"])
    x1, c1 = eng.debug_stage(src["blocks"], carrier_offset=rec["carrier_offset"])
    car = (rec["flags"] & F.FLAG_CARRIER) != 0
    assert np.array_equal(x0[car], x1[car]) and np.array_equal(c0[car], c1[car])
    x2, _ = eng.debug_stage(src["blocks"], carrier_offset=np.zeros(len(rec)))
    assert not np.array_equal(x0[car], x2[car])
    with pytest.raises(ValueError):
        eng.debug_stage(src["blocks"], carrier_offset=np.zeros(3))


def test_an_exception_of_the_callable_belongs_to_its_block_and_the_modes_it_excludes(golden):
    src = golden("c2")
    st = settings_of(src)
    calls = []

    def picky(corr_mag, peak):
        calls.append(peak)
        if len(calls) == 4:
            raise FloatingPointError("no vertex")
        return onp.xcorr_parabolic(corr_mag, peak)

    det = Detector(st, iter(items_of(src)), rxid=0, batch_size=8)
    det.soa_estimate.interpolate = picky
    out = []
    with pytest.raises(FloatingPointError, match="no vertex"):
        for item in det:
            out.append(item)
    hits = np.flatnonzero(src["det"])
    assert len(out) == hits[3]                       # every block before the fourth detection came out
    with pytest.raises(StopIteration):
        next(det)
    det2 = Detector(st, io.BytesIO(b""), rxid=0)
    det2.soa_estimate.interpolate = onp.xcorr_none
    with pytest.raises(TypeError, match="replaced interpolator"):
        next(det2.iter_detected_records())
    with pytest.raises(NotImplementedError):
        det2.soa_estimate(np.zeros(16384, np.complex64))
    from thrifty_amd.experimental.detect_preshift import PreshiftDetector
    pre = PreshiftDetector(st, None)
    with pytest.raises(NotImplementedError):
        pre.soa_estimate.interpolate = onp.xcorr_none
    with pytest.raises(TypeError):
        Detector(st, None).soa_estimate.interpolate = 0.25
    with pytest.raises(NotImplementedError):
        Detector(st, None).soa_estimate.interpolate(np.ones(8), 3)   # the engine's own has no host form
