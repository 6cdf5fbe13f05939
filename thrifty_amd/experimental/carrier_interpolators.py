"""Carrier peak interpolators selectable for the experimental detectors
(reference thrifty/experimental/carrier_interpolators.py).

For `PreshiftDetector` the interpolation happens inside the detection kernel (`preshift_verdict()`
in csrc/detect16k_preshift.hip, float32 like the magnitudes it is given) and the three-point
functions here are the *selectors* the reference API passes around -- `PreshiftDetector(...,
interpolator=gaussian)`.  For the default `Detector` any of them -- the curve-fitting ones
(`make_dirichlet`, `make_parabole_fit`, `make_corr_parabolic`) included -- can be assigned to
`Detector.sync.interpolator` like in the reference (experimental/detect_carrier_interpol.py) and is
then EVALUATED here, on the host, between two engine passes (thrifty_amd.detect: the slow path).
"""
import numpy as np


def _dirichlet_kernel(xdata, block_len, carrier_len):
    """sin(pi W x / N) / (W sin(pi x / N)), 1 at x = 0 (reference carrier_interpolators.py:7-14)."""
    x = np.array(xdata, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        weights = np.sin(np.pi * carrier_len * x / block_len) / np.sin(np.pi * x / block_len) / carrier_len
        weights[np.isnan(weights)] = 1
    return weights


def none(fft_mag, peak):
    """No sub-bin estimate (reference carrier_interpolators.py:17-18)."""
    return 0


def parabolic(fft_mag, peak):
    """Sub-bin carrier offset from a parabola through |X[peak-1]|, |X[peak]|, |X[peak+1]|
    (reference carrier_interpolators.py:40-45)."""
    left, mid, right = fft_mag[peak - 1], fft_mag[peak], fft_mag[peak + 1]
    return (right - left) / (4 * mid - 2 * left - 2 * right)


def gaussian(fft_mag, peak):
    """The same on the logarithms of the three magnitudes (reference :48-54)."""
    left, mid, right = np.log(fft_mag[peak - 1]), np.log(fft_mag[peak]), np.log(fft_mag[peak + 1])
    return (right - left) / (4 * mid - 2 * left - 2 * right)


def cosine(fft_mag, peak):
    """Cosine fit through the three magnitudes (reference :84-92)."""
    left, mid, right = fft_mag[peak - 1], fft_mag[peak], fft_mag[peak + 1]
    cos_omega = (left + right) / (2 * mid)
    if cos_omega > 1:
        return 0
    omega = np.arccos(cos_omega)
    theta = np.arctan((left - right) / (2 * mid * np.sin(omega)))
    return -theta / omega


def make_dirichlet(block_len, carrier_len, width=6):
    """Least-squares fit of A |Dirichlet(x - offset)| to the `width + 1` magnitudes around the peak
    (reference :21-37; SciPy's curve_fit, as there)."""
    from scipy.optimize import curve_fit

    def _fit_model(xdata, amplitude, time_offset):
        x = np.array(xdata, dtype=np.float64)
        return amplitude * np.abs(_dirichlet_kernel(x - time_offset, block_len, carrier_len))

    def _interpolator(fft_mag, peak):
        xdata = np.arange(-(width // 2), width // 2 + 1)
        ydata = fft_mag[peak + xdata]
        popt, _ = curve_fit(_fit_model, xdata, ydata, p0=(fft_mag[peak], 0))
        return popt[1]

    return _interpolator


def make_parabole_fit(width):
    """Vertex of the least-squares parabola through `width + 1` magnitudes (reference :57-66)."""
    def _interpolator(fft_mag, peak):
        xdata = np.arange(-(width // 2), width // 2 + 1)
        coeffs = np.polyfit(xdata, fft_mag[peak + xdata], 2)
        return -coeffs[1] / coeffs[0] / 2

    return _interpolator


def make_corr_parabolic(corr_width, block_len, carrier_len):
    """Three-point parabola on the magnitudes correlated with the Dirichlet kernel (reference :69-81)."""
    rel = np.arange(-(corr_width // 2), corr_width // 2 + 1)
    dirichlet = _dirichlet_kernel(rel, block_len, carrier_len)

    def _interpolator(fft_mag, peak):
        left = np.sum(fft_mag[peak + rel - 1] * dirichlet)
        mid = np.sum(fft_mag[peak + rel] * dirichlet)
        right = np.sum(fft_mag[peak + rel + 1] * dirichlet)
        return (right - left) / (4 * mid - 2 * left - 2 * right)

    return _interpolator


INTERPOLATORS = {"none": none, "parabolic": parabolic, "gaussian": gaussian, "cosine": cosine}
