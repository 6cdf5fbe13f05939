"""A replaced carrier interpolator on the GPU engine: `Detector.sync.interpolator = fn`, what the
reference's InterpolationDetector does (thrifty/experimental/detect_carrier_interpol.py:17-40 with
the functions of carrier_interpolators.py:17-81).  The callable runs on the host between two engine
passes (thr_detect_offsets).  Fixtures `interpol_c2_*` come from running the reference class; the
callables assigned here are the ORACLE's restatements (oracle/thrifty_np.py) -- the product ships the
hook, not the reference's interpolators."""
import io

import numpy as np
import pytest

from thrifty_amd import _native as F
from thrifty_amd import block_data
from thrifty_amd.detect import Detector
from oracle import thrifty_np as onp

from test_gpu_detector_api import card_text, settings_of

pytestmark = pytest.mark.gpu

METHODS = {"none": lambda st: onp.no_offset, "parabolic": lambda st: onp.parabolic_offset,
           "gaussian": lambda st: onp.gaussian_offset, "cosine": lambda st: onp.cosine_offset,
           "parabole_fit6": lambda st: onp.parabole_fit_offset(6),
           "corr_parabolic4": lambda st: onp.corr_parabolic_offset(4, st.block_len, st.carrier_len)}


def host_dirichlet(st):
    """The engine's own fit restated on the host (carrier_sync.py:150-196: curve_fit on seven magnitudes)."""
    return lambda mag, peak: onp.dirichlet_fit(mag, peak, st.block_len, st.carrier_len)[1]


@pytest.mark.parametrize("name", sorted(METHODS))
def test_interpolation_detector_matches_the_references(golden, name):
    g = golden("interpol_c2_" + name)
    src = golden(str(g["src"]))
    st = settings_of(src)
    items = [(1000.0 + i, int(src["block_idx"][i]), src["blocks"][i]) for i in range(len(src["blocks"]))]
    det = Detector(st, iter(items), rxid=int(src["rxid"]), batch_size=7)
    det.sync.interpolator = METHODS[name](st)
    assert det._host_interp
    got = list(det)
    assert len(got) == len(items)
    lines = []
    for i, (detected, res) in enumerate(got):
        assert res.carrier_info.bin == g["cbin"][i]
        assert (res.corr_info is not None) == bool(g["carrier_det"][i]) and detected == bool(g["det"][i])
        np.testing.assert_allclose(res.carrier_info.energy, g["cenergy"][i], rtol=1e-5)
        np.testing.assert_allclose(res.carrier_info.noise, g["cnoise"][i], rtol=1e-4)
        if res.corr_info is None:
            continue
        # the offset is the CALLABLE's value and type: none() and cosine's early return give the int 0
        assert isinstance(res.carrier_info.offset, int) == bool(g["coff_is_int"][i])
        # (float32 magnitudes of another FFT: 5e-5 bin -- relative where the interpolator itself is
        # ill-conditioned: corr_parabolic's denominator nearly cancels on some blocks, offsets of -4 bins)
        np.testing.assert_allclose(float(res.carrier_info.offset), g["coff"][i], atol=5e-5, rtol=2e-4)
        assert res.corr_info.sample == g["sample"][i]                       # bit-exact SoA sample
        np.testing.assert_allclose(res.corr_info.energy, g["energy"][i], rtol=1e-4)
        np.testing.assert_allclose(res.corr_info.noise, g["noise"][i], rtol=1e-4)
        if detected:
            np.testing.assert_allclose(res.corr_info.offset, g["soff"][i], atol=1e-4)
            np.testing.assert_allclose(res.soa, g["soa"][i], atol=2e-4)
            lines.append(res.serialize())
    want = str(g["toad"]).split("\n")
    assert len(lines) == len(want)
    for a, b in zip(lines, want):
        fa, fb = a.split(), b.split()
        assert fa[:3] == fb[:3] and fa[4] == fb[4] and fa[8] == fb[8]       # rxid ts block | sample | bin
        if fb[9] == "0":
            assert fa[9] == "0"                                            # the int 0 prints as "0"


def test_the_engines_own_fit_restated_on_the_host_gives_the_engines_records(golden):
    """SciPy's curve_fit on seven magnitudes assigned as the interpolator is the default detector
    computed the slow way: same bins, samples and verdicts, offsets to the tolerance of the device fit."""
    g = golden("c2")
    st = settings_of(g)
    items = [(1000.0 + i, int(g["block_idx"][i]), g["blocks"][i]) for i in range(len(g["blocks"]))]
    fast = Detector(st, iter(items), rxid=0)
    assert not fast._host_interp
    want = list(fast)
    slow = Detector(st, iter(items), rxid=0, batch_size=5)
    slow.sync.interpolator = host_dirichlet(st)
    assert slow._host_interp and slow.sync.interpolator is not None
    got = list(slow)
    assert len(got) == len(want)
    for (d1, r1), (d2, r2) in zip(want, got):
        assert d1 == d2 and r1.carrier_info.bin == r2.carrier_info.bin
        if r1.corr_info is None:
            assert r2.corr_info is None
            continue
        np.testing.assert_allclose(r2.carrier_info.offset, r1.carrier_info.offset, atol=2e-4)
        assert r2.corr_info.sample == r1.corr_info.sample
        np.testing.assert_allclose(r2.corr_info.energy, r1.corr_info.energy, rtol=2e-5)
    # None = no sub-bin estimate (carrier_sync.py:66-68): the reference's `none`, offset 0
    off = Detector(st, iter(items), rxid=0)
    off.sync.interpolator = None
    g0 = golden("interpol_c2_none")
    for i, (detected, res) in enumerate(off):
        assert detected == bool(g0["det"][i])
        if res.corr_info is not None:
            assert res.carrier_info.offset == 0 and res.corr_info.sample == g0["sample"][i]


def test_an_exception_of_the_callable_belongs_to_its_block_and_file_readers_still_work(golden, tmp_path):
    g = golden("c2")
    st = settings_of(g)
    path = tmp_path / "rx.card"
    path.write_text(card_text(g))
    calls = []

    def picky(fft_mag, peak):
        calls.append(peak)
        assert fft_mag.dtype == np.float32 and fft_mag.shape == (16384,)
        if len(calls) == 6:
            raise IndexError("index 16385 is out of bounds for axis 0 with size 16384")
        return onp.parabolic_offset(fft_mag, peak)

    with open(path, "rb") as f:
        det = Detector(st, block_data.CardStream(f, 16384), rxid=0, batch_size=4)    # a mapped file, device ingest ...
        assert det._pin
        det.sync.interpolator = picky                                                 # ... until the slow path takes over
        assert not det._pin and det._card is None
        out = []
        with pytest.raises(IndexError, match="16385"):
            for item in det:
                out.append(item)
        with pytest.raises(StopIteration):
            next(det)
    carriers = np.flatnonzero(g["carrier_det"])
    assert len(out) == carriers[5]                  # every block before the sixth carrier-positive one came out
    assert [res.block for _, res in out] == [int(b) for b in g["block_idx"][:len(out)]]
    # record iteration and the library loop are not offered in this mode
    det2 = Detector(st, io.BytesIO(b""), rxid=0)
    det2.sync.interpolator = onp.no_offset
    with pytest.raises(TypeError, match="replaced interpolator"):
        next(det2.iter_detected_records())
    with pytest.raises(NotImplementedError):
        det2.sync(g["blocks"][0])
    # the variants interpolate inside their fused kernels
    from thrifty_amd.experimental.detect_preshift import PreshiftDetector
    with pytest.raises(NotImplementedError):
        PreshiftDetector(st, None).sync.interpolator = onp.no_offset
    with pytest.raises(TypeError):
        Detector(st, None).sync.interpolator = 3


def test_detect_offsets_entry_point(golden):
    """thr_detect_offsets directly: zero offsets == interpolator `none`; the fit's own offsets fed
    back reproduce thr_detect's records bit for bit; long and short blocks take it too."""
    g = golden("c2")
    eng = F.Engine(16384, 4096, g["template"], (0, 15, 0), (7, 110), (0, 15, 0), max_batch=32)
    rec = eng.detect(g["blocks"], g["block_idx"])[:, 0]
    again = eng.detect_offsets(g["blocks"], rec["carrier_offset"], g["block_idx"])[:, 0]
    ok = (rec["flags"] & F.FLAG_INDEX_ERROR) == 0
    assert again[ok].tobytes() == rec[ok].tobytes()
    zero = eng.detect_offsets(g["blocks"], np.zeros(len(rec)), g["block_idx"])[:, 0]
    g0 = golden("interpol_c2_none")
    car = g0["carrier_det"]
    assert np.array_equal(zero["corr_sample"][car], g0["sample"][car]) and np.all(zero["carrier_offset"] == 0)
    with pytest.raises(F.NativeError, match="default detector only"):
        F.Engine(16384, 4096, g["template"], (0, 15, 0), (7, 110), (0, 15, 0), preshift_num=21).detect_offsets(
            g["blocks"][:2], np.zeros(2))
    for name in ("c3", "small"):
        gg = golden(name)
        e = F.Engine(int(gg["block_len"]), int(gg["history_len"]), gg["template"], tuple(gg["carrier_thresh"]),
                     tuple(int(v) for v in gg["carrier_window"]), tuple(gg["corr_thresh"]), max_batch=32)
        r = e.detect(gg["blocks"], gg["block_idx"])[:, 0]
        a = e.detect_offsets(gg["blocks"], r["carrier_offset"], gg["block_idx"])[:, 0]
        keep = (r["flags"] & F.FLAG_INDEX_ERROR) == 0
        assert a[keep].tobytes() == r[keep].tobytes(), name


def test_a_non_finite_offset_of_the_callable_is_the_blocks_error(golden):
    """A callable that returns nan / inf (gaussian on a zero magnitude: log(0)): the reference's
    shifter raises on that block (int(round(nan)), carrier_sync.py:241-245); here the blocks before
    it come out, then ValueError -- and thr_detect_offsets itself refuses such an array."""
    g = golden("c2")
    st = settings_of(g)
    items = [(1000.0 + i, int(g["block_idx"][i]), g["blocks"][i]) for i in range(len(g["blocks"]))]
    calls = []

    def broken(fft_mag, peak):
        calls.append(peak)
        return float("nan") if len(calls) == 4 else 0.25

    det = Detector(st, iter(items), rxid=0, batch_size=6)
    det.sync.interpolator = broken
    out = []
    with pytest.raises(ValueError, match="not finite"):
        for item in det:
            out.append(item)
    carriers = np.flatnonzero(g["carrier_det"])
    assert len(out) == carriers[3]
    eng = F.Engine(16384, 4096, g["template"], (0, 15, 0), (7, 110), (0, 15, 0), max_batch=8)
    with pytest.raises(F.NativeError, match="not finite"):
        eng.detect_offsets(g["blocks"][:3], np.array([0.0, np.inf, 0.0]))
