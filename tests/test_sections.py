"""The overlap-save plan of the long-block correlate stage (thr_plan_sections, csrc/handle.hip:
plan_sections), checked on the CPU against the oracle's `despread` / `corr_peak`
(reference soa_estimator.py:97-102, 137-143): sectioning must reproduce the reference's kept
lags, its windowed first-max, the peak's neighbours and the stddev sums -- exactly the claims
detect_seg.hip builds on.  No GPU involved: the planner is a host-only entry point of the C ABI."""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import synth

M = 16384

GEOMETRIES = [
    # block_len, history_len, template_len
    (65536, 4096, 4094),      # BASELINE configs[2]
    (32768, 4096, 4094),
    (65536, 4096, 1023),
    (65536, 9400, 9361),      # the longest template that still sections at 65536
    (32768, 5000, 4914),
    (65536, 4096, 2),
    (65536, 65000, 3),
]


def _plan(n, h, w):
    return F.plan_sections(n, h, w)


@pytest.mark.parametrize("n,h,w", GEOMETRIES)
def test_plan_tiles_window_and_lags_exactly_once(n, h, w):
    secs = _plan(n, h, w)
    assert 2 <= len(secs) <= 8
    corr_len = n - w + 1
    lo, hi = onp.unique_window(n, h, w)
    v = M - w + 1                                       # valid lags of a section: [0, v)
    win_seen = np.zeros(corr_len, dtype=int)
    sum_seen = np.zeros(corr_len, dtype=int)
    prev_hi = 0
    for g, s in enumerate(secs):
        assert s["start"] % 2 == 0                      # u8 samples are fetched as 4-byte pairs
        assert 0 <= s["start"] and s["start"] + M <= n  # a section never leaves the block
        assert s["sum_lo"] == prev_hi                   # ascending, gap-free
        prev_hi = s["sum_hi"]
        assert s["start"] <= s["sum_lo"] and s["sum_hi"] <= s["start"] + v   # only exact lags
        sum_seen[s["sum_lo"]:s["sum_hi"]] += 1
        if s["win_hi"] > s["win_lo"]:
            win_seen[s["win_lo"]:s["win_hi"]] += 1
            assert s["sum_lo"] <= s["win_lo"] and s["win_hi"] <= s["sum_hi"]
            # both neighbours of every searched lag are exact lags of the same section (the block's
            # own first / last lag has no neighbour in the reference either: soa_estimator.py:160-161)
            assert s["win_lo"] - 1 >= s["start"] or s["win_lo"] == 0
            assert s["win_hi"] <= s["start"] + v - 1 or s["win_hi"] == corr_len
    assert prev_hi == corr_len
    assert np.all(sum_seen == 1)
    expect = np.zeros(corr_len, dtype=int)
    expect[lo:hi] = 1
    assert np.array_equal(win_seen, expect)


@pytest.mark.parametrize("n,h,w", [(16384, 4920, 4914), (16384, 1100, 1023), (65536, 9400, 9362), (65536, 20000, 16384),
                                   (8192, 2048, 511), (131072, 12000, 12000)])
def test_geometries_that_are_not_sectioned(n, h, w):
    assert _plan(n, h, w) == []


def test_bad_geometry_is_refused():
    with pytest.raises(F.NativeError):
        F.plan_sections(65536, 100, 4094)       # history < template_len - 1 (soa_estimator.py:32)
    with pytest.raises(F.NativeError):
        F.plan_sections(60000, 4096, 4094)      # not a power of two


def _sectioned_stats(y, tpl, secs):
    """What detect_seg.hip + k_finish compute, in float64 NumPy: per section one 16384-point
    circular correlation, the first-max over its window lags, then the first best section."""
    t16 = np.conj(np.fft.fft(np.concatenate([tpl, np.zeros(M - len(tpl))])))
    best = None
    s1 = s2 = 0.0
    for s in secs:
        c = np.fft.ifft(np.fft.fft(y[s["start"]:s["start"] + M]) * t16)
        mag = np.abs(c)
        a, b = s["sum_lo"] - s["start"], s["sum_hi"] - s["start"]
        s1 += mag[a:b].sum()
        s2 += (mag[a:b] ** 2).sum()
        a, b = s["win_lo"] - s["start"], s["win_hi"] - s["start"]
        if b <= a:
            continue
        j = int(np.argmax(mag[a:b])) + a
        cand = (mag[j], s["start"] + j, mag[j - 1] if j > 0 else None, mag[j + 1])
        if best is None or cand[0] > best[0]:
            best = cand
    return best, s1, s2


@pytest.mark.parametrize("n,h,bits,sps", [(65536, 4096, 11, 2.0), (32768, 4096, 11, 2.0),
                                           (65536, 4096, 10, 1.0), (65536, 8400, 11, 4.1)])
def test_sectioned_correlation_equals_the_oracle_despread(n, h, bits, sps):
    tpl = synth.gold_template(bits, 3, sps).astype(np.float64)
    w = len(tpl)
    secs = _plan(n, h, w)
    assert secs
    bank = onp.TemplateBank(tpl, n, h)
    rng = np.random.default_rng(n + w)
    blocks, _ = synth.synth_blocks(rng, 3, n, tpl, bank.window, signal_frac=0.67)
    for raw in blocks:
        x = onp.iq_u8_to_c64(raw).astype(np.complex128)
        shift = -(37 + 0.3137)
        y = x * np.exp(2j * np.pi * shift * (np.arange(n) / n - 0.5))     # carrier_sync.py:222-238
        corr = onp.despread(np.fft.fft(y), bank)
        mag = np.abs(corr)
        idx, peak = onp.corr_peak(mag, bank.window)
        best, s1, s2 = _sectioned_stats(y, tpl, secs)
        assert best[1] == idx
        np.testing.assert_allclose(best[0], peak, rtol=1e-12)
        np.testing.assert_allclose(best[2], mag[idx - 1], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(best[3], mag[idx + 1], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(s1, mag.sum(), rtol=1e-12)
        np.testing.assert_allclose(s2, (mag ** 2).sum(), rtol=1e-12)


def test_first_max_tie_across_sections_takes_the_lower_lag():
    """np.argmax returns the first maximum (soa_estimator.py:139); two equal peaks in different
    sections must resolve to the earlier section."""
    n, h = 65536, 4096
    tpl = synth.gold_template(11, 3, 2.0).astype(np.float64)
    secs = _plan(n, h, len(tpl))
    y = np.zeros(n, dtype=np.complex128)
    p0, p1 = 5000, 40000                       # sections 0 and 3
    y[p0:p0 + len(tpl)] = tpl
    y[p1:p1 + len(tpl)] = tpl
    bank = onp.TemplateBank(tpl, n, h)
    mag = np.abs(onp.despread(np.fft.fft(y), bank))
    idx, _ = onp.corr_peak(mag, bank.window)
    best, _, _ = _sectioned_stats(y, tpl, secs)
    # (float64 rounding may make one of the two a hair larger; both agree on which)
    assert best[1] == idx
    assert idx in (p0, p1)
