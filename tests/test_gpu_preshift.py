"""GPU parity of the PreshiftDetector variant (SURVEY.md 8(f) rank 2;
csrc/detect16k_preshift.hip + the multi-pass pipeline for other block lengths) against
fixtures produced by the reference's thrifty/experimental/detect_preshift.py and against the
pinned oracle restatement."""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import block_data, synth
from thrifty_amd.detect import DetectorSettings
from thrifty_amd.experimental.detect_preshift import PreshiftDetector

pytestmark = pytest.mark.gpu

from conftest import golden_blocks

FIXTURES = ["preshift_c2", "preshift_c2_straddle", "preshift_c2_stddev", "preshift_c1", "preshift_small",
            # the reference's other three-point carrier interpolators (carrier_interpolators.py)
            "preshift_c2_none", "preshift_c2_gaussian", "preshift_c2_cosine", "preshift_c2_straddle_gaussian"]


def interp_of(g):
    return str(g["interpolator"]) if "interpolator" in g.files else "parabolic"


def engine_for(g, **kw):
    return F.Engine(int(g["block_len"]), int(g["history_len"]), g["template"],
                    tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                    tuple(g["corr_thresh"]), preshift_num=int(g["num"]), interpolator=interp_of(g), **kw)


def check_records(rec, g):
    """bit-exact bin / sample index / verdicts; offsets and energies to the stated tolerance."""
    n = len(golden_blocks(g))
    assert len(rec) == n
    for i in range(n):
        r = rec[i]
        assert r["block_idx"] == g["block_idx"][i]
        assert r["carrier_bin"] == g["cbin"][i]
        if g["index_error"][i]:
            assert r["flags"] & F.FLAG_INDEX_ERROR and not r["flags"] & F.FLAG_CARRIER
            continue
        assert bool(r["flags"] & F.FLAG_CARRIER) == bool(g["carrier_det"][i])
        assert bool(r["flags"] & F.FLAG_CORR) == bool(g["det"][i])
        np.testing.assert_allclose(r["carrier_energy"], g["cenergy"][i], rtol=1e-5)
        np.testing.assert_allclose(r["carrier_noise"], g["cnoise"][i], rtol=1e-4)
        if not g["carrier_det"][i]:
            continue
        np.testing.assert_allclose(r["carrier_offset"], g["coff"][i], atol=2e-5)
        assert r["corr_sample"] == g["sample"][i]                       # bit-exact SoA sample
        np.testing.assert_allclose(r["corr_energy"], g["energy"][i], rtol=1e-4)
        np.testing.assert_allclose(r["corr_noise"], g["noise"][i], rtol=1e-4)
        if g["det"][i]:
            np.testing.assert_allclose(r["corr_offset"], g["soff"][i], atol=1e-4)


@pytest.mark.parametrize("name", FIXTURES)
def test_matches_reference_goldens(golden, name):
    g = golden(name)
    rec = engine_for(g, max_batch=8).detect(golden_blocks(g), g["block_idx"])[:, 0]
    check_records(rec, g)
    # the debug word carries the roll and the bank index the reference would have used
    for i in range(len(rec)):
        if g["carrier_det"][i]:
            shift = -(float(g["cbin"][i]) + float(g["coff"][i]))
            assert np.int32(np.uint32(rec[i]["reserved"] >> np.uint64(32))) == int(np.round(shift))
            frac = g["frac_shift"][i]
            assert int(rec[i]["reserved"] & np.uint64(0xFFFFFFFF)) == int(np.round((frac + 0.5) * (int(g["num"]) - 1)))


def test_c64_input_and_generic_pipeline_agree(golden):
    g = golden("preshift_c2")
    eng = engine_for(g, max_batch=8)
    rec_u8 = eng.detect(g["blocks"], g["block_idx"])[:, 0]
    c64 = np.stack([block_data.raw_to_complex(b) for b in g["blocks"]])
    rec_c = eng.detect(c64, g["block_idx"])[:, 0]
    assert rec_u8.tobytes() == rec_c.tobytes()
    eng_gen = engine_for(g, max_batch=8, path="multipass")
    rec_gen = eng_gen.detect(g["blocks"], g["block_idx"])[:, 0]
    check_records(rec_gen, g)
    assert np.array_equal(rec_gen["corr_sample"], rec_u8["corr_sample"])
    assert np.array_equal(rec_gen["reserved"], rec_u8["reserved"])


@pytest.mark.parametrize("num", [1, 2, 21, 64])
def test_random_blocks_match_oracle(num):
    n, h = 16384, 4096
    tpl = synth.gold_template(10, 3, 1.0)
    win = onp.unique_window(n, h, len(tpl))
    rng = np.random.default_rng(1000 + num)
    nb = 300
    blocks, _ = synth.synth_blocks(rng, nb, n, tpl, win, signal_frac=0.85)
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=128, preshift_num=num)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    orc = onp.OraclePreshiftDetector(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), num=num)
    hits = bank_flips = 0
    for i in range(nb):
        res = orc.detect_u8(i, blocks[i])
        r = rec[i]
        assert r["carrier_bin"] == res.carrier.bin
        assert bool(r["flags"] & F.FLAG_CARRIER) == res.carrier.detected
        if not res.carrier.detected:
            continue
        # float32 parabola: |X| differs from np.abs(complex64) by an ulp or two and the
        # denominator 4b - 2a - 2c amplifies that; SURVEY 8(c) allows 1e-3 bin here
        np.testing.assert_allclose(r["carrier_offset"], res.carrier.offset, atol=1e-4)
        if int(r["reserved"] & np.uint64(0xFFFFFFFF)) != orc.last[2]:
            bank_flips += 1          # offset within float rounding of a bank boundary
            continue
        assert bool(r["flags"] & F.FLAG_CORR) == res.detected
        assert r["corr_sample"] == res.corr.sample
        np.testing.assert_allclose(r["corr_energy"], res.corr.energy, rtol=1e-4)
        if res.detected:
            hits += 1
            np.testing.assert_allclose(r["corr_offset"], res.corr.offset, atol=1e-4)
    assert hits > 200 and bank_flips <= 1


@pytest.mark.parametrize("name", ["preshift_c2", "preshift_c2_none", "preshift_c2_gaussian", "preshift_c2_cosine"])
def test_detector_class_serialises_like_the_reference(golden, name):
    from thrifty_amd.experimental import carrier_interpolators
    g = golden(name)
    gb = golden_blocks(g)
    st = DetectorSettings(int(g["block_len"]), int(g["history_len"]), len(g["template"]),
                          tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                          g["template"], tuple(g["corr_thresh"]))
    blocks = [(1000.0 + i, int(g["block_idx"][i]), gb[i]) for i in range(len(gb))]
    # the reference passes the interpolator FUNCTION (detect_preshift.py:48-58)
    det = PreshiftDetector(st, blocks, rxid=int(g["rxid"]), num=int(g["num"]), batch_size=7,
                           interpolator=carrier_interpolators.INTERPOLATORS[interp_of(g)])
    lines = [res.serialize() for detected, res in det if detected]
    # ... and the engine-formatted text (thr_format_toad) is the same text
    det2 = PreshiftDetector(st, blocks, rxid=int(g["rxid"]), num=int(g["num"]), batch_size=7,
                            interpolator=interp_of(g))
    assert [ln for chunk in det2.iter_toad_lines() for ln in chunk] == lines
    want = str(g["toad"]).split("\n")
    assert len(lines) == len(want)
    for a, b in zip(lines, want):
        fa, fb = a.split(), b.split()
        assert fa[:3] == fb[:3] and fa[4] == fb[4] and fa[8] == fb[8]      # rxid ts block | sample | bin
        np.testing.assert_allclose(float(fa[3]), float(fb[3]), atol=2e-4)   # soa
        np.testing.assert_allclose([float(v) for v in fa[5:8]], [float(v) for v in fb[5:8]], rtol=1e-4, atol=1e-4)
        if interp_of(g) == "none":
            assert fa[9] == fb[9] == "0"                                   # none() returns the int 0
            continue
        np.testing.assert_allclose(float(fa[9]), float(fb[9]), atol=2e-5)   # float32 carrier offset
        assert float(np.float32(float(fa[9]))) == float(fa[9])   # a widened float32, like the reference's
    with pytest.raises(NotImplementedError):
        PreshiftDetector(st, None, interpolator=lambda m, p: 0)
    with pytest.raises(F.NativeError, match="exactly one template"):
        F.Engine(16384, 4096, np.ones((2, 100)), (0, 15, 0), (7, 110), (0, 15, 0), preshift_num=21)


def test_cosine_interpolator_returns_the_int_zero_where_the_reference_does():
    """carrier_interpolators.py:84-92: `if cos_omega > 1: return 0` -- an int, so that block's
    carrier-offset column reads "0" (not "0.0") and CarrierSyncInfo.offset is the int 0.  It takes
    a neighbour larger than the windowed peak: a carrier just BELOW the window's first bin puts the
    window's maximum on that bin with the stronger bin outside."""
    from thrifty_amd.experimental import carrier_interpolators
    n, h = 16384, 4096
    tpl = synth.gold_template(10, 2)
    win = onp.unique_window(n, h, len(tpl))
    rng = np.random.default_rng(77)
    # a continuous carrier at bin 6.15 .. 6.32 under the burst, the window starting at 7: the window's
    # maximum is bin 7 with |X[6]| well above it -- cos(omega) = 1.25 .. 3.4; at 6.41 / 6.45 it is
    # 0.74 .. 0.87 and the formula applies.  (A burst alone has a main lobe 16 bins wide: its three
    # magnitudes are nearly equal and cos(omega) sits within rounding of 1.)
    cars = [6.15, 6.19, 6.24, 6.28, 6.32, 6.41, 6.45] * 2
    blocks = []
    for car in cars:
        p = int(rng.integers(win[0], win[1]))
        z = rng.normal(0, 0.02, n) + 1j * rng.normal(0, 0.02, n)
        z += 0.08 * np.exp(2j * np.pi * car * np.arange(n) / n)
        k = np.arange(len(tpl))
        z[p:p + len(tpl)] += 0.3 * (tpl + 1) / 2 * np.exp(2j * np.pi * car * (k + p) / n)
        blocks.append(synth.quantise_iq(z))
    blocks = np.stack(blocks)
    st = DetectorSettings(n, h, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
    items = [(1000.0 + i, i, blocks[i]) for i in range(len(blocks))]
    det = PreshiftDetector(st, items, rxid=0, num=21, batch_size=5, interpolator=carrier_interpolators.cosine)
    got = list(det)
    orc = onp.OraclePreshiftDetector(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), num=21, interpolator="cosine")
    ints = 0
    for i, (detected, res) in enumerate(got):
        want = orc.detect_u8(i, blocks[i])
        assert res.carrier_info.bin == want.carrier.bin and (res.corr_info is not None) == want.carrier.detected
        if not want.carrier.detected:
            continue
        if isinstance(want.carrier.offset, int):       # the reference's `return 0`
            ints += 1
            assert res.carrier_info.offset == 0 and isinstance(res.carrier_info.offset, int)
            assert detected == want.detected
            if detected:
                assert res.serialize().split()[9] == "0" == onp.toad_line(0, 1000.0 + i, i, want).split()[9]
        else:
            assert isinstance(res.carrier_info.offset, np.floating)
    assert ints == 10
    # the library's own formatter (thr_format_toad) and the column formatter print the same text
    det2 = PreshiftDetector(st, items, rxid=0, num=21, batch_size=5, interpolator="cosine")
    lines = [ln for chunk in det2.iter_toad_lines() for ln in chunk]
    assert lines == [res.serialize() for detected, res in got if detected]
    from thrifty_amd import toads_data
    det3 = PreshiftDetector(st, items, rxid=0, num=21, batch_size=50, interpolator="cosine")
    for stamps, recs in det3.iter_detected_records():
        assert toads_data.toad_lines(recs, stamps, n - h, rxid=0, carrier_offset_type=np.float32) == lines


def test_index_error_is_raised_like_the_reference(golden):
    g = golden("preshift_c2_straddle")
    st = DetectorSettings(int(g["block_len"]), int(g["history_len"]), len(g["template"]),
                          tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                          g["template"], tuple(g["corr_thresh"]))
    det = PreshiftDetector(st, None, num=int(g["num"]))
    bad = int(np.flatnonzero(g["index_error"])[0])
    with pytest.raises(IndexError):
        det.detect(0.0, 0, g["blocks"][bad])


@pytest.mark.parametrize("name,n_check", [("preshift_c2", 6), ("preshift_small", 6)])
def test_yield_data_returns_the_rolled_spectrum_and_the_correlation(golden, name, n_check):
    """Reference: PreshiftDetector(..., yield_data=True).detect() -> (detected, result, rolled
    FFT#1, correlation) (detect.py:60-78 with detect_preshift.py:62-80).  Served by the multi-pass
    kernels; records equal the fused kernel's, dumps equal the oracle's intermediates."""
    g = golden(name)
    n, h = int(g["block_len"]), int(g["history_len"])
    st = DetectorSettings(n, h, len(g["template"]), tuple(g["carrier_thresh"]),
                          tuple(int(v) for v in g["carrier_window"]), g["template"], tuple(g["corr_thresh"]))
    det = PreshiftDetector(st, None, rxid=0, yield_data=True, num=int(g["num"]))
    plain = PreshiftDetector(st, None, rxid=0, num=int(g["num"]))
    orc = onp.OraclePreshiftDetector(n, h, g["template"], tuple(g["carrier_thresh"]),
                                     tuple(int(v) for v in g["carrier_window"]), tuple(g["corr_thresh"]),
                                     num=int(g["num"]))
    seen = 0
    for i in range(len(g["blocks"])):
        if g["index_error"][i]:
            continue
        detected, res, rolled, corr = det.detect(1.0, int(g["block_idx"][i]), g["blocks"][i])
        d2, r2 = plain.detect(1.0, int(g["block_idx"][i]), g["blocks"][i])
        assert detected == d2 == bool(g["det"][i])
        assert res.carrier_info.bin == r2.carrier_info.bin == g["cbin"][i]
        if not g["carrier_det"][i]:
            assert rolled is None and corr is None and res.corr_info is None
            continue
        assert res.corr_info.sample == r2.corr_info.sample == g["sample"][i]
        np.testing.assert_allclose(res.corr_info.energy, r2.corr_info.energy, rtol=2e-5)
        _, (orolled, ocorr) = orc.detect_u8(int(g["block_idx"][i]), g["blocks"][i], want_data=True)
        assert rolled.shape == (n,) and corr.shape == ocorr.shape == (n - len(g["template"]) + 1,)
        assert np.linalg.norm(rolled - orolled) / np.linalg.norm(orolled) < 2e-6
        assert np.linalg.norm(corr - ocorr) / np.linalg.norm(ocorr) < 1e-5
        lo, hi = det.soa_estimate.window
        assert int(np.argmax(np.abs(corr[lo:hi]))) + lo == res.corr_info.sample
        seen += 1
        if seen >= n_check:
            break
    assert seen >= 4
