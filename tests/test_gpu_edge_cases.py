"""GPU edge cases through the C ABI: empty and ragged batches, degenerate inputs,
noise-only batches (empty work list), saturated samples -- HIP engine vs oracle."""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import synth

pytestmark = pytest.mark.gpu

N, H = 16384, 4096
TPL = synth.gold_template(10, 2)
WIN = onp.unique_window(N, H, len(TPL))


def make_engine(max_batch=64, window=(7, 110), cthr=(0, 15, 0)):
    return F.Engine(N, H, TPL, cthr, window, (0, 15, 0), max_batch=max_batch)


def oracle(window=(7, 110), cthr=(0, 15, 0)):
    return onp.OracleDetector(N, H, TPL, cthr, window, (0, 15, 0))


def compare(rec, blocks, orc, idx=None):
    for i, raw in enumerate(blocks):
        (res,) = orc.detect_u8(i if idx is None else int(idx[i]), raw)
        r = rec[i]
        assert r["carrier_bin"] == res.carrier.bin, i
        assert bool(r["flags"] & F.FLAG_CARRIER) == res.carrier.detected, i
        np.testing.assert_allclose(r["carrier_energy"], res.carrier.energy, rtol=1e-4, atol=1e-6)
        if np.isfinite(res.carrier.noise):
            np.testing.assert_allclose(r["carrier_noise"], res.carrier.noise, rtol=1e-4, atol=1e-6)
        if res.carrier.detected:
            assert r["corr_sample"] == res.corr.sample, i
            assert bool(r["flags"] & F.FLAG_CORR) == res.corr.detected, i
            np.testing.assert_allclose(r["corr_energy"], res.corr.energy, rtol=1e-4)
            np.testing.assert_allclose(r["corr_offset"], res.corr.offset, atol=1e-4)


def test_empty_batch_is_a_noop():
    eng = make_engine()
    out = eng.detect(np.zeros((0, 2 * N), dtype=np.uint8))
    assert out.shape == (0, 1)


@pytest.mark.parametrize("nb", [1, 3, 257, 300])
def test_ragged_batches_cover_every_block(nb):
    """Block counts that are not multiples of the grid (256 CUs) or of max_batch."""
    rng = np.random.default_rng(nb)
    seed_blocks, _ = synth.synth_blocks(rng, 8, N, TPL, WIN)
    blocks = seed_blocks[np.arange(nb) % 8]
    idx = np.arange(nb) * 5 + 11
    eng = make_engine(max_batch=128)
    rec = eng.detect(blocks, idx)[:, 0]
    assert np.array_equal(rec["block_idx"], idx)
    ref = make_engine(max_batch=8).detect(seed_blocks)[:, 0]
    for i in range(nb):
        for f in ("flags", "carrier_bin", "corr_sample", "corr_energy", "corr_offset", "carrier_offset"):
            assert rec[i][f] == ref[i % 8][f], (i, f)          # bit-identical regardless of batching


def test_noise_only_batch_has_empty_work_list():
    rng = np.random.default_rng(5)
    blocks, _ = synth.synth_blocks(rng, 20, N, TPL, WIN, signal_frac=0.0)
    rec = make_engine().detect(blocks)[:, 0]
    assert not np.any(rec["flags"] & (F.FLAG_CARRIER | F.FLAG_CORR))
    assert np.all(rec["corr_sample"] == -1)
    compare(rec, blocks, oracle())
    # ... and the engine still works afterwards (work counter re-armed)
    sig, _ = synth.synth_blocks(rng, 4, N, TPL, WIN)
    eng = make_engine()
    eng.detect(blocks)
    compare(eng.detect(sig)[:, 0], sig, oracle())


def test_degenerate_inputs_match_oracle():
    consts = [np.full(2 * N, v, dtype=np.uint8) for v in (0, 127, 128, 255)]
    nyq = np.tile(np.array([0, 0, 255, 255], dtype=np.uint8), N // 2)        # Nyquist tone
    iq_only = np.tile(np.array([255, 127], dtype=np.uint8), N)               # I saturated, Q mid
    rng = np.random.default_rng(9)
    sat = rng.integers(0, 2, 2 * N).astype(np.uint8) * 255                   # random 0/255
    blocks = np.stack(consts + [nyq, iq_only, sat])
    compare(make_engine().detect(blocks)[:, 0], blocks, oracle())


def test_strong_signal_near_full_scale():
    rng = np.random.default_rng(12)
    blocks, _ = synth.synth_blocks(rng, 6, N, TPL, WIN, amp=0.95, sigma=0.01)
    compare(make_engine().detect(blocks)[:, 0], blocks, oracle())


def test_constant_threshold_only(golden):
    """thresh = (c, 0, 0): detection decided purely by the constant term."""
    g = golden("c2")
    cthr = (40.0 ** 2, 0.0, 0.0)
    eng = F.Engine(N, H, g["template"], cthr, (7, 110), (150.0 ** 2, 0.0, 0.0), max_batch=32)
    orc = onp.OracleDetector(N, H, g["template"], cthr, (7, 110), (150.0 ** 2, 0.0, 0.0))
    compare(eng.detect(g["blocks"])[:, 0], g["blocks"], orc)


@pytest.mark.parametrize("n,bits,sps", [
    (64, 4, 1.0),            # the smallest block the engine accepts (template 15 chips)
    (1 << 20, 12, 8.0),      # the largest: 1 Mi samples per block, 32 760-sample template
])
def test_block_length_limits(n, bits, sps):
    """thr_create's documented range [64, 2^20] at both ends, against the oracle."""
    # (the Gold tables cover 5..11 bits: build +-1 codes of the wanted length directly)
    tpl = np.repeat(np.sign(np.random.default_rng(bits).normal(0, 1, (1 << bits) - 1)), int(sps))
    h = len(tpl) + 8
    win = onp.unique_window(n, h, len(tpl))
    rng = np.random.default_rng(n)
    lo, hi = (3.2, 12.0) if n == 64 else (10.0, 100.0)
    cwin = (2, 14) if n == 64 else (7, 110)
    blocks, _ = synth.synth_blocks(rng, 3, n, tpl, win, signal_frac=1.0, carrier_bins=(lo, hi))
    eng = F.Engine(n, h, tpl, (0, 8, 0), cwin, (0, 8, 0), max_batch=2)
    rec = eng.detect(blocks, np.arange(3))[:, 0]
    orc = onp.OracleDetector(n, h, tpl, (0, 8, 0), cwin, (0, 8, 0))
    hits = 0
    for i in range(3):
        try:
            (res,) = orc.detect_u8(i, blocks[i])
        except IndexError:
            assert rec[i]["flags"] & F.FLAG_INDEX_ERROR
            continue
        assert rec[i]["carrier_bin"] == res.carrier.bin
        assert bool(rec[i]["flags"] & F.FLAG_CARRIER) == res.carrier.detected
        if res.carrier.detected:
            assert rec[i]["corr_sample"] == res.corr.sample
            assert bool(rec[i]["flags"] & F.FLAG_CORR) == res.corr.detected
            np.testing.assert_allclose(rec[i]["corr_energy"], res.corr.energy, rtol=2e-4)
            hits += res.corr.detected
    assert hits >= 1
    with pytest.raises(F.NativeError):
        F.Engine(n * 2 if n > 64 else 32, 8, np.ones(4), (0, 8, 0), (1, 3), (0, 8, 0))
