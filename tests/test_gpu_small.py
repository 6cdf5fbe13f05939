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
def test_lds_path_agrees_with_the_multi_pass_pipeline(n, monkeypatch):
    h, tpl, cwin, blocks, _ = make_case(n, 90, seed=77 + n)
    thr = (0, 15, 0)
    fast = F.Engine(n, h, tpl, thr, cwin, thr, max_batch=32).detect(blocks)[:, 0]
    monkeypatch.setenv("THR_FORCE_GENERIC", "1")
    slow_eng = F.Engine(n, h, tpl, thr, cwin, thr, max_batch=32)
    monkeypatch.delenv("THR_FORCE_GENERIC")
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
    # yield_data intermediates come from the multi-pass pipeline (allocated on demand)
    xhat, corr = both.debug_stage(blocks[:3], template_id=0)
    lo, hi = onp.unique_window(n, h, len(tpl))
    for i in range(3):
        if rec[i, 0]["flags"] & F.FLAG_CARRIER:
            assert int(np.argmax(np.abs(corr[i][lo:hi]))) + lo == rec[i, 0]["corr_sample"]


def test_stddev_terms_fall_back_to_the_multi_pass_pipeline(golden):
    """(the LDS kernels carry no stddev sums: such settings still work, through generic.hip)"""
    n = 4096
    h, tpl, cwin, blocks, _ = make_case(n, 24, seed=6)
    eng = F.Engine(n, h, tpl, (0, 12, 1.0), cwin, (0, 12, 0.5), max_batch=16)
    rec = eng.detect(blocks)[:, 0]
    rows = soak_util.run_oracle(blocks, n, h, tpl, (0, 12, 1.0), cwin, (0, 12, 0.5), procs=4, chunk=8)
    mism, worst, ties = soak_util.compare(rec, rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR)
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst)
