"""The host-side interpolators of thrifty_amd.experimental (what an analysis script assigns to
`Detector.sync.interpolator` / `Detector.soa_estimate.interpolate`) against the oracle's restatement
of the reference's (oracle/thrifty_np.py, pinned to the reference by the `interpol_*` / `xcorr_*`
fixtures in test_oracle_golden.py).  CPU only: no engine involved."""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd.experimental import carrier_interpolators, xcorr_interpolators


def peaky(rng, n, at, width, dtype=np.float32):
    x = np.arange(n)
    return (np.exp(-0.5 * ((x - at) / width) ** 2) * 50 + rng.random(n) * 0.5 + 0.1).astype(dtype)


@pytest.mark.parametrize("name", ["none", "parabolic", "gaussian", "cosine"])
def test_three_point_correlation_interpolators(name):
    rng = np.random.default_rng(3)
    fn, want = xcorr_interpolators.INTERPOLATORS[name], getattr(onp, "xcorr_" + name)
    for k in range(50):
        at = 20 + 10 * rng.random()
        mag = peaky(rng, 64, at, 0.6 + rng.random())
        peak = int(np.argmax(mag))
        a, b = fn(mag, peak), want(mag, peak)
        assert type(a) is type(b)
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7)
    flat = np.array([1.0, 3.0, 1.0, 7.0], dtype=np.float32)     # (a + c) / 2b > 1: cosine has no solution
    assert xcorr_interpolators.cosine(flat, 2) == 0 and isinstance(xcorr_interpolators.cosine(flat, 2), int)
    assert isinstance(xcorr_interpolators.none(flat, 1), int)


def test_autocorr_fit_and_maximise():
    rng = np.random.default_rng(5)
    chips = rng.integers(0, 2, 127) * 2.0 - 1.0
    template = np.repeat(chips, 4)                               # a float template, oversampled like c1's
    n = 2048
    x = np.zeros(n, np.complex64)
    x[300:300 + len(template)] = template
    shift = 0.3
    spec = np.fft.fft(x) * np.exp(-2j * np.pi * shift * np.fft.fftfreq(n))
    x = (np.fft.ifft(spec) + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    xhat = np.fft.fft(x)
    bank = onp.TemplateBank(template, n, len(template) + 40)
    corr = onp.despread(xhat, bank)
    mag = np.abs(corr)
    peak = int(np.argmax(mag))
    assert peak == 300
    a = xcorr_interpolators.make_autocorr_fit(template)(mag, peak)
    np.testing.assert_allclose(a, onp.xcorr_autocorr(template)(mag, peak), atol=1e-6)
    assert -0.55 <= a <= 0.55
    guess = xcorr_interpolators.gaussian(mag, peak)
    m = xcorr_interpolators.make_maximise(template)(np.fft.ifft(xhat), peak, guess)
    np.testing.assert_allclose(m, onp.xcorr_maximise(template)(mag, peak, xhat), atol=1e-6)
    assert abs(m - shift) < 0.05                                 # it does find the delay
    with pytest.raises(TypeError):                               # an integer template: NumPy refuses the in-place scale
        xcorr_interpolators.make_autocorr_fit(chips.astype(np.int64))(mag, peak)
    assert sorted(xcorr_interpolators.INTERPOLATORS) == ["autocorr", "cosine", "gaussian", "maximise", "none",
                                                         "parabolic"]


@pytest.mark.parametrize("name,ref", [("none", "no_offset"), ("parabolic", "parabolic_offset"),
                                      ("gaussian", "gaussian_offset"), ("cosine", "cosine_offset")])
def test_three_point_carrier_interpolators(name, ref):
    rng = np.random.default_rng(7)
    fn, want = carrier_interpolators.INTERPOLATORS[name], getattr(onp, ref)
    for k in range(50):
        mag = peaky(rng, 64, 20 + 10 * rng.random(), 0.6 + rng.random())
        peak = int(np.argmax(mag))
        a, b = fn(mag, peak), want(mag, peak)
        assert type(a) is type(b)
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7)
