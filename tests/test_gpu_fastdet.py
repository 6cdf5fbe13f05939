"""GPU tests of the fastdet-compatible variant (`thr_create_fastdet`).
PARITY UNPINNED against the reference: fastdet cannot be built here (FFTW3f / VOLK /
librtlsdr); the checks are GPU == the NumPy restatement of cardet.c + corr_detector.cpp
(oracle.OracleFastdet), plus the structural facts of the algorithm."""
import io

import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import block_data, fastdet, synth
from thrifty_amd.detect import DetectorSettings

pytestmark = pytest.mark.gpu

# power-domain thresholds: |X|^2 > c + s * noise_power
CTHR, XTHR = (0.0, 200.0, 0.0), (0.0, 200.0, 0.0)


def compare(rec, res, nb):
    hits = 0
    for i in range(nb):
        r, w = rec[i], res[i]
        assert r["carrier_bin"] == w.argmax
        assert bool(r["flags"] & F.FLAG_CARRIER) == w.carrier
        np.testing.assert_allclose(r["carrier_energy"], np.sqrt(w.carrier_max), rtol=1e-5)
        np.testing.assert_allclose(r["carrier_noise"], np.sqrt(w.carrier_noise), rtol=1e-4)
        if not w.carrier:
            continue
        np.testing.assert_allclose(r["carrier_offset"], w.carrier_offset, atol=1e-4)
        assert abs(r["carrier_offset"]) <= 0.5
        assert r["corr_sample"] == w.peak_idx                       # bit-exact sample index
        assert bool(r["flags"] & F.FLAG_CORR) == w.detected
        np.testing.assert_allclose(r["corr_energy"], np.sqrt(w.peak_power), rtol=1e-4)
        np.testing.assert_allclose(r["corr_noise"], np.sqrt(w.noise_power), rtol=1e-3, atol=1e-3)
        if w.detected:
            hits += 1
            np.testing.assert_allclose(r["corr_offset"], w.peak_offset, atol=1e-4)
            assert abs(r["corr_offset"]) <= 0.5
    return hits


@pytest.mark.parametrize("n,h,bits,window", [
    (16384, 4096, 10, (7, 110)),        # fused kernel
    (16384, 4096, 10, (-110, -7)),      # negative bins (both ends), fused kernel
    (4096, 1024, 9, (5, 60)),           # multi-pass pipeline
])
def test_gpu_matches_restatement(n, h, bits, window):
    tpl = synth.gold_template(bits, 2, 1.0).astype(np.float32)
    win = onp.unique_window(n, h, len(tpl))
    rng = np.random.default_rng(n + window[0])
    nb = 200
    bins = (-100.0, -10.0) if window[0] < 0 else (10.0, min(100.0, window[1] - 5.0))
    blocks, _ = synth.synth_blocks(rng, nb, n, tpl, win, signal_frac=0.8, carrier_bins=bins)
    eng = F.Engine(n, h, tpl, CTHR, window, XTHR, max_batch=64, fastdet=True)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    orc = onp.OracleFastdet(n, h, tpl, CTHR, window, XTHR)
    res = [orc.detect_u8(i, blocks[i]) for i in range(nb)]
    assert compare(rec, res, nb) > 100


def test_refusals():
    tpl = synth.gold_template(10, 2, 1.0)
    with pytest.raises(F.NativeError, match="window range not supported"):
        F.Engine(16384, 4096, tpl, CTHR, (-12, 12), XTHR, fastdet=True)      # cardet.c:44-48
    with pytest.raises(F.NativeError, match="no stddev term"):
        F.Engine(16384, 4096, tpl, (0, 15, 1.0), (7, 110), XTHR, fastdet=True)
    with pytest.raises(ValueError):
        onp.fastdet_window(-12, 12, 16384)


def test_fastdetector_lines_and_tpl(tmp_path):
    n, h = 16384, 4096
    tpl = synth.gold_template(10, 2, 1.0).astype(np.float32)
    fastdet.save_tpl(tmp_path / "t.tpl", tpl)
    assert np.array_equal(fastdet.load_tpl(tmp_path / "t.tpl"), tpl)
    win = onp.unique_window(n, h, len(tpl))
    blocks, _ = synth.synth_blocks(np.random.default_rng(4), 12, n, tpl, win, signal_frac=0.75)
    text = "".join(block_data.card_line(1475000000.25 + i, 40 + i, blocks[i]) for i in range(12))
    (tmp_path / "rx.card").write_text(text)
    (tmp_path / "detector.cfg").write_text(
        "rxid: 2\nsample_rate: 2.4M\nblock_size: %d\nblock_history: %d\ncarrier_window: 7 - 110\n"
        "carrier_threshold: 200 * snr\ncorr_threshold: 200*snr\n" % (n, h))
    fastdet._main([str(tmp_path / "rx.card"), "-o", str(tmp_path / "rx.toad"), "--tpl",
                   str(tmp_path / "t.tpl"), "-c", str(tmp_path / "detector.cfg")])
    got = (tmp_path / "rx.toad").read_text().strip().split("\n")
    orc = onp.OracleFastdet(n, h, tpl, CTHR, (7, 110), XTHR)
    want = []
    for i in range(12):
        r = orc.detect_u8(40 + i, blocks[i])
        if r.detected:
            want.append(onp.fastdet_toad_line(2, 1475000000.25 + i, 40 + i, r))
    assert len(got) == len(want) >= 6
    for a, b in zip(got, want):
        fa, fb = a.split(), b.split()
        assert fa[:3] == fb[:3] and fa[4] == fb[4] and fa[8] == fb[8]
        assert len(fa) == 12 and len(fa[5].split(".")[1]) == 12            # %.12f offset
        np.testing.assert_allclose([float(v) for v in fa[3:]], [float(v) for v in fb[3:]], rtol=1e-4, atol=2e-4)
