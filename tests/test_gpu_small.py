"""GPU parity of the LDS-resident short-block kernels (csrc/detect_small.hip): block_len 1024,
2048, 4096 and 8192 -- the lengths the reference's own tests use
(tests/test_carrier_detect.py:57 runs 8192).  Against the reference-generated fixture
(`small`, N = 4096), against the CPU oracle on fresh seeded blocks (batch sizes that leave the
last group of 16 / R1 blocks partly filled), and against the multi-pass pipeline."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_util  # noqa: E402
from oracle import thrifty_np as onp  # noqa: E402
from thrifty_amd import _native as F  # noqa: E402
from thrifty_amd import synth  # noqa: E402
from test_gpu_parity import check_against_golden, engine_for  # noqa: E402

pytestmark = pytest.mark.gpu

# block_len -> (history, Gold register bits of the template, carrier window)
GEOMETRY = {1024: (256, 7, (3, 60)), 2048: (512, 8, (5, 100)), 4096: (1024, 9, (7, 110)),
            8192: (2048, 10, (7, 110))}


def make_case(n, nb, seed, signal_frac=0.8):
    h, bits, cwin = GEOMETRY[n]
    tpl = synth.gold_template(bits, 2).astype(np.float64)
    win = onp.unique_window(n, h, len(tpl))
    rng = np.random.default_rng(seed)
    hi = min(cwin[1] - 5.0, n / 8.0)
    blocks, truth = synth.synth_blocks(rng, nb, n, tpl, win, signal_frac=signal_frac,
                                       carrier_bins=(cwin[0] + 3.0, hi))
    return h, tpl, cwin, blocks, truth


def test_small_fixture_from_the_reference(golden):
    g = golden("small")
    assert int(g["block_len"]) == 4096
    for mb in (3, 64):      # 3: every launch is one partly filled group
        rec = engine_for(g, max_batch=mb).detect(g["blocks"], g["block_idx"])
        check_against_golden(rec[:, 0], g)


@pytest.mark.parametrize("n,nb", [(1024, 203), (2048, 150), (4096, 131), (8192, 77)])
def test_fresh_short_blocks_equal_the_oracle(n, nb):
    h, tpl, cwin, blocks, truth = make_case(n, nb, seed=900 + n)
    thr = (0, 15, 0)
    eng = F.Engine(n, h, tpl, thr, cwin, thr, max_batch=64)
    idx = np.arange(nb) * 3 + 1
    rec = eng.detect(blocks, idx)[:, 0]
    assert np.array_equal(rec["block_idx"], idx)
    rows = soak_util.run_oracle(blocks, n, h, tpl, thr, cwin, thr, procs=8, chunk=32)
    mism, worst, ties = soak_util.compare(rec, rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR,
                                          only=np.asarray(truth["has_signal"], dtype=bool))
    assert sum(1 for r in rows if r is not None and r[5]) > 0.4 * nb
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst, ties)
    assert not ties
    assert worst["offset"] <= 5e-6 and worst["energy"] <= 2e-5 and worst["noise"] <= 2e-5, worst
    # (short templates: the Dirichlet lobe is wide against the 7 fitted bins -- DESIGN.md section 4)
    assert worst["car_off"] <= (2e-4 if n >= 4096 else 2e-3), worst
    # complex64 input (a block that was already converted) gives the same records
    rec2 = eng.detect(np.stack([onp.iq_u8_to_c64(b) for b in blocks]), idx)[:, 0]
    assert rec2.tobytes() == rec.tobytes()


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_lds_path_agrees_with_the_multi_pass_pipeline(n):
    h, tpl, cwin, blocks, _ = make_case(n, 90, seed=77 + n)
    thr = (0, 15, 0)
    fast = F.Engine(n, h, tpl, thr, cwin, thr, max_batch=32).detect(blocks)[:, 0]
    slow_eng = F.Engine(n, h, tpl, thr, cwin, thr, max_batch=32, path="multipass")
    slow = slow_eng.detect(blocks)[:, 0]
    assert np.array_equal(fast["carrier_bin"], slow["carrier_bin"])
    assert np.array_equal(fast["corr_sample"], slow["corr_sample"])
    assert np.array_equal(fast["flags"], slow["flags"])
    np.testing.assert_allclose(fast["corr_energy"], slow["corr_energy"], rtol=2e-5)
    np.testing.assert_allclose(fast["corr_offset"], slow["corr_offset"], atol=2e-5)
    np.testing.assert_allclose(fast["carrier_energy"], slow["carrier_energy"], rtol=2e-5)


def test_short_blocks_two_templates_and_stage_dumps():
    n = 4096
    h, tpl, cwin, blocks, _ = make_case(n, 40, seed=5)
    tpl2 = synth.gold_template(9, 5).astype(np.float64)
    thr = (0, 15, 0)
    both = F.Engine(n, h, np.stack([tpl, tpl2]), thr, cwin, thr, max_batch=16)
    rec = both.detect(blocks)
    one = F.Engine(n, h, tpl, thr, cwin, thr, max_batch=16).detect(blocks)[:, 0]
    two = F.Engine(n, h, tpl2, thr, cwin, thr, max_batch=16).detect(blocks)[:, 0]
    for got, want in ((rec[:, 0], one), (rec[:, 1], two)):
        assert np.array_equal(got["corr_sample"], want["corr_sample"])
        assert np.array_equal(got["flags"], want["flags"])
        np.testing.assert_allclose(got["corr_energy"], want["corr_energy"], rtol=1e-6)
    assert np.all(rec[:, 1]["template_id"] == 1)
    xhat, corr = both.debug_stage(blocks[:3], template_id=0)
    lo, hi = onp.unique_window(n, h, len(tpl))
    for i in range(3):
        if rec[i, 0]["flags"] & F.FLAG_CARRIER:
            assert int(np.argmax(np.abs(corr[i][lo:hi]))) + lo == rec[i, 0]["corr_sample"]


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_stage_dumps_of_the_lds_kernels_equal_the_oracle(n):
    """Detector.detect(yield_data=True)'s intermediates (detect.py:75-78) and the FFT test hook now
    come from the LDS-resident kernels' dump mode: spectrum vs np.fft, shifted spectrum and
    correlation vs the oracle, for either template of a two-template engine; the records of the
    dump call equal the plain call's."""
    h, tpl, cwin, blocks, _ = make_case(n, 21, seed=40 + n, signal_frac=1.0)
    bits = GEOMETRY[n][1]
    tpl2 = synth.gold_template(bits, 5).astype(np.float64)
    thr = (0, 15, 0)
    eng = F.Engine(n, h, np.stack([tpl, tpl2]), thr, cwin, thr, max_batch=32)
    spec = eng.debug_fft(blocks[:19])            # 19: a partly filled last group for every R1
    for i in range(19):
        ref = np.fft.fft(onp.iq_u8_to_c64(blocks[i]).astype(np.complex128))
        assert np.linalg.norm(spec[i] - ref) / np.linalg.norm(ref) < 2e-6
    plain = eng.detect(blocks[:19])
    for t, tp in enumerate((tpl, tpl2)):
        orc = onp.OracleDetector(n, h, tp, thr, cwin, thr)
        xhat, corr = eng.debug_stage(blocks[:19], template_id=t)
        for i in range(19):
            (res,), ((xh, co),) = orc.detect_u8(0, blocks[i], want_data=True)
            assert res.carrier.detected
            # (n < 4096: the short template's Dirichlet lobe is wide against the 7 fitted bins, the
            # fitted carrier offset -- and with it the shift -- is only good to ~1e-3 bins, see
            # test_fresh_short_blocks_equal_the_oracle; a shift error d bins is a phase ramp of
            # +-pi d across the block)
            tol = 5e-6 if n >= 4096 else 3e-3
            assert np.linalg.norm(xhat[i] - xh) / np.linalg.norm(xh) < tol
            assert np.linalg.norm(corr[i][:len(co)] - co) / np.linalg.norm(co) < tol
            assert plain[i, t]["corr_sample"] == res.corr.sample
    # complex64 input through the same mode
    spec2 = eng.debug_fft(np.stack([onp.iq_u8_to_c64(b) for b in blocks[:5]]))
    np.testing.assert_allclose(spec2, spec[:5], rtol=0, atol=1e-4 * np.abs(spec[:5]).max())


@pytest.mark.parametrize("n,nb", [(1024, 150), (2048, 120), (4096, 100), (8192, 60)])
def test_stddev_terms_run_in_the_lds_kernels_and_equal_the_oracle(n, nb):
    """Thresholds with a stddev term (carrier_detect.py:110-115, soa_estimator.py:127-134): the
    short-block kernels carry the sums themselves now (no detour through the multi-pass
    pipeline); records against the oracle on fresh blocks, and against the multi-pass pipeline."""
    h, tpl, cwin, blocks, truth = make_case(n, nb, seed=6 + n)
    cthr, xthr = (0, 12, 1.0), (0, 12, 0.5)
    eng = F.Engine(n, h, tpl, cthr, (0, -1), xthr, max_batch=64)       # full window: every |X| counts
    rec = eng.detect(blocks)[:, 0]
    rows = soak_util.run_oracle(blocks, n, h, tpl, cthr, (0, -1), xthr, procs=8, chunk=16)
    mism, worst, ties = soak_util.compare(rec, rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR,
                                          only=np.asarray(truth["has_signal"], dtype=bool))
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst, ties)
    assert len(ties) <= 1
    assert worst["offset"] <= 5e-6 and worst["energy"] <= 2e-5 and worst["noise"] <= 2e-5, worst
    slow_eng = F.Engine(n, h, tpl, cthr, (0, -1), xthr, max_batch=64, path="multipass")
    slow = slow_eng.detect(blocks)[:, 0]
    assert np.array_equal(rec["flags"], slow["flags"]) and np.array_equal(rec["corr_sample"], slow["corr_sample"])
    # a verdict that flips with the stddev coefficient proves the term is live in these kernels
    hard = F.Engine(n, h, tpl, (0, 12, 1e6), (0, -1), xthr, max_batch=64).detect(blocks)[:, 0]
    assert ((rec["flags"] & F.FLAG_CARRIER) != 0).sum() > 0.4 * nb
    assert ((hard["flags"] & F.FLAG_CARRIER) != 0).sum() == 0
    hard2 = F.Engine(n, h, tpl, cthr, (0, -1), (0, 12, 1e9), max_batch=64).detect(blocks)[:, 0]
    assert ((hard2["flags"] & F.FLAG_CORR) != 0).sum() == 0 < ((rec["flags"] & F.FLAG_CORR) != 0).sum()
