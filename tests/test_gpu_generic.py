"""GPU parity of the generic-length pipeline (csrc/generic.hip): BASELINE config C3
(N = 65536), the small-block fixture (N = 4096, also the .card -> .toad text case), and
the N = 16384 fixtures forced through the generic path as a cross-check of the fast one."""
import io
import os

import numpy as np
import pytest

from thrifty_amd import _native, block_data
from thrifty_amd.detect import Detector, DetectorSettings

from test_gpu_parity import check_against_golden, engine_for
from test_gpu_detector_api import assert_toad_close, settings_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["small", "c3"])
def test_generic_lengths_match_reference_golden(golden, name):
    g = golden(name)
    eng = engine_for(g, max_batch=8)     # also exercises internal sub-batching
    rec = eng.detect(g["blocks"], g["block_idx"])
    check_against_golden(rec[:, 0], g)
    spec = eng.debug_fft(g["blocks"][:2])
    for i in range(2):
        ref = np.fft.fft(block_data.raw_to_complex(g["blocks"][i]).astype(np.complex128))
        assert np.linalg.norm(spec[i] - ref) / np.linalg.norm(ref) < 2e-6


@pytest.mark.parametrize("name", ["c2", "c2_stddev", "c2_negwin"])
def test_generic_path_agrees_with_fast_path(golden, name):
    g = golden(name)
    fast = engine_for(g).detect(g["blocks"], g["block_idx"])[:, 0]
    slow_eng = engine_for(g, path="multipass")
    slow = slow_eng.detect(g["blocks"], g["block_idx"])[:, 0]
    check_against_golden(slow, g)
    assert np.array_equal(fast["carrier_bin"], slow["carrier_bin"])
    assert np.array_equal(fast["corr_sample"], slow["corr_sample"])
    assert np.array_equal(fast["flags"], slow["flags"])
    np.testing.assert_allclose(fast["corr_energy"], slow["corr_energy"], rtol=2e-5)
    np.testing.assert_allclose(fast["corr_offset"], slow["corr_offset"], atol=2e-5)


def test_small_card_stream_to_toad(golden):
    """The reference's own card_reader -> Detector -> serialize chain on a .card text."""
    g = golden("small")
    det = Detector(settings_of(g), block_data.card_reader(io.StringIO(str(g["card_text"]))),
                   rxid=3, batch_size=4)
    lines = [res.serialize() for detected, res in det if detected]
    assert_toad_close(lines, g["card_toad"])


def test_generic_multi_template_and_dumps(golden):
    gs = [golden("c5_tx%d" % i) for i in range(4)]
    tpls = np.stack([g["template"] for g in gs]).astype(np.float64)
    eng = engine_for(gs[0], templates=tpls, path="multipass")
    rec = eng.detect(gs[0]["blocks"], gs[0]["block_idx"])
    for t, g in enumerate(gs):
        check_against_golden(rec[:, t], g)
    xhat, corr = eng.debug_stage(gs[0]["blocks"][:2], template_id=2)
    for i in range(2):
        assert int(np.argmax(np.abs(corr[i][1537:13825]))) + 1537 == rec[i, 2]["corr_sample"]
        np.testing.assert_allclose(np.mean(np.abs(xhat[i]) ** 2), gs[0]["xhat_energy"][i], rtol=1e-5)
