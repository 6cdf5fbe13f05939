"""The overlap-save plan of a 16384-sample block with a short template (thr_plan_sections at
block_len 16384, csrc/handle.hip: plan_sections_4k): up to four sections of 4096 samples cover the
unique window, which is what csrc/detect16k_sec.hip builds on.  Checked on the CPU against the
oracle's `despread` / `corr_peak` (reference soa_estimator.py:97-102, 137-143, 159-170): the
sections' windowed first-max, its power and the peak's two neighbours must be the block's.  No GPU
involved: the planner is a host-only entry point of the C ABI."""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import synth

N, M = 16384, 4096


def test_baseline_geometry_is_four_sections_of_3072_lags():
    secs = F.plan_sections(N, 4096, 1023)
    assert [s["start"] for s in secs] == [1536, 4608, 7680, 10752]
    assert [(s["win_lo"], s["win_hi"]) for s in secs] == [(1537, 4609), (4609, 7681), (7681, 10753), (10753, 13825)]
    # in section coordinates every section owns [1, 3073): rows 0 .. 2 of 1024 lags but lag 0, and lag 3072
    assert all((s["win_lo"] - s["start"], s["win_hi"] - s["start"]) == (1, 3073) for s in secs)


def _check_plan(h, w, secs):
    lo, hi = onp.unique_window(N, h, w)
    corr_len = N - w + 1
    v = M - w + 1
    assert 1 <= len(secs) <= 4
    seen = np.zeros(corr_len, dtype=int)
    prev = lo
    for s in secs:
        assert s["start"] % 8 == 0                          # 16-byte sample fetches per thread
        assert 0 <= s["start"] and s["start"] + M <= N      # a section never leaves the block
        assert s["win_lo"] == prev and s["win_hi"] > s["win_lo"]   # ascending, gap-free
        prev = s["win_hi"]
        assert s["start"] <= s["win_lo"] and s["win_hi"] <= s["start"] + v   # only exact lags
        # both neighbours of every owned lag are exact lags of the same section -- but for the block's
        # own first / last kept lag, where the reference takes none (soa_estimator.py:163-164)
        assert s["win_lo"] - 1 >= s["start"] or s["win_lo"] == 0
        assert s["win_hi"] <= s["start"] + v - 1 or s["win_hi"] == corr_len
        seen[s["win_lo"]:s["win_hi"]] += 1
    expect = np.zeros(corr_len, dtype=int)
    expect[lo:hi] = 1
    assert np.array_equal(seen, expect)


def test_plans_tile_the_unique_window_exactly_once():
    rng = np.random.default_rng(7)
    planned = refused = 0
    cases = [(4096, 1023), (5120, 1023), (8200, 1023), (2300, 200), (6200, 1200), (1022, 1023), (16383, 2),
             (4094, 4081), (12300, 30), (8190, 1025), (4096, 1025), (4095, 1024)]
    cases += [(int(h), int(w)) for w in rng.integers(2, 4200, 300) for h in [rng.integers(w - 1, N)]]
    for h, w in cases:
        secs = F.plan_sections(N, h, w)
        lo, hi = onp.unique_window(N, h, w)
        if hi <= lo:
            continue
        if secs:
            _check_plan(h, w, secs)
            planned += 1
        else:
            # refused only when four sections cannot hold the window: a section owns at most
            # 4096 - w - 1 lags, and one that starts on a multiple of 8 may lose up to 7 of them
            refused += 1
            assert w > M - 15 or hi - lo > 4 * (M - w - 1 - 7) - 8, (h, w)
    assert planned > 60 and refused > 60
    assert F.plan_sections(N, 4096, 1023) and not F.plan_sections(N, 1100, 1023)
    assert not F.plan_sections(N, 4920, 4914)      # the example detector.cfg: template longer than a section


def _sectioned_peak(y, tpl, secs):
    """What detect16k_sec.hip + k_finish compute, in float64 NumPy: per section one 4096-point
    circular correlation, the first-max over its owned lags, then the first best section."""
    t4 = np.conj(np.fft.fft(np.concatenate([tpl, np.zeros(M - len(tpl))])))
    best = None
    for s in secs:
        mag = np.abs(np.fft.ifft(np.fft.fft(y[s["start"]:s["start"] + M]) * t4))
        a, b = s["win_lo"] - s["start"], s["win_hi"] - s["start"]
        j = int(np.argmax(mag[a:b])) + a
        cand = (mag[j], s["start"] + j, mag[j - 1] if j > 0 else None, mag[j + 1])
        if best is None or cand[0] > best[0]:
            best = cand
    return best


@pytest.mark.parametrize("h,w", [(4096, 1023), (5120, 1023), (8200, 1023), (2300, 200), (6200, 1200), (12300, 30)])
def test_sectioned_correlation_equals_the_oracle_despread(h, w):
    rng = np.random.default_rng(h + w)
    tpl = synth.gold_template(10, 3).astype(np.float64) if w == 1023 else np.sign(rng.normal(0, 1, w))
    secs = F.plan_sections(N, h, w)
    assert secs
    bank = onp.TemplateBank(tpl, N, h)
    lo, hi = bank.window
    # bursts on the first and last lag of the window and on both sides of every section boundary
    edges = [lo, lo + 1, hi - 1, hi - 2] + [p for s in secs[1:] for p in (s["win_lo"] - 1, s["win_lo"], s["win_lo"] + 1)]
    pos = np.array(edges + list(rng.integers(lo, hi, 6)))
    blocks, _ = synth.synth_blocks(rng, len(pos), N, tpl, bank.window, positions=pos)
    for raw, p in zip(blocks, pos):
        x = onp.iq_u8_to_c64(raw).astype(np.complex128)
        shift = -(41 + 0.2718)
        y = x * np.exp(2j * np.pi * shift * (np.arange(N) / N - 0.5))     # carrier_sync.py:222-238
        mag = np.abs(onp.despread(np.fft.fft(y), bank))
        idx, peak = onp.corr_peak(mag, bank.window)
        best = _sectioned_peak(y, tpl, secs)
        assert best[1] == idx
        np.testing.assert_allclose(best[0], peak, rtol=1e-12)
        if idx > 0:
            np.testing.assert_allclose(best[2], mag[idx - 1], rtol=1e-9, atol=1e-12)
        if idx + 1 < len(mag):
            np.testing.assert_allclose(best[3], mag[idx + 1], rtol=1e-9, atol=1e-12)


def test_equal_peaks_in_two_sections():
    """Two equal bursts in different sections: in exact arithmetic a tie, which np.argmax resolves to
    the lower lag (soa_estimator.py:139) and k_finish to the earlier section (strict '>' over
    ascending sections).  In floating point the two transforms round differently, so either lag may
    come out a hair larger -- the sectioned and the unsectioned form must still agree on the peak's
    magnitude to rounding, and both must name one of the two."""
    tpl = synth.gold_template(10, 3).astype(np.float64)
    secs = F.plan_sections(N, 4096, len(tpl))
    y = np.zeros(N, dtype=np.complex128)
    p0, p1 = 2000, 11000                       # sections 0 and 3
    y[p0:p0 + len(tpl)] = tpl
    y[p1:p1 + len(tpl)] = tpl
    bank = onp.TemplateBank(tpl, N, 4096)
    mag = np.abs(onp.despread(np.fft.fft(y), bank))
    idx, peak = onp.corr_peak(mag, bank.window)
    best = _sectioned_peak(y, tpl, secs)
    assert idx in (p0, p1) and best[1] in (p0, p1)
    np.testing.assert_allclose(best[0], peak, rtol=1e-12)
    np.testing.assert_allclose(mag[p0], mag[p1], rtol=1e-12)
