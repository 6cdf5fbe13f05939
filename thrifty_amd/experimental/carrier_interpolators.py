"""Carrier peak interpolators selectable for the experimental detectors
(reference thrifty/experimental/carrier_interpolators.py).

On the GPU the interpolation happens inside the detection kernel (`preshift_verdict()` in
csrc/detect16k_preshift.hip, float32 like the magnitudes it is given); the functions here are the
*selectors* the reference API passes around -- `PreshiftDetector(..., interpolator=gaussian)` --
plus a host evaluation of the same three-point formulas for analysis scripts.  The reference's
curve-fitting interpolators (`make_dirichlet`, `make_parabole_fit`, `make_corr_parabolic`) have no
device form in this variant: the Dirichlet fit is what the DEFAULT `Detector` runs.
"""
import numpy as np


def none(fft_mag, peak):
    """No sub-bin estimate (reference carrier_interpolators.py:17-18)."""
    return 0


def parabolic(fft_mag, peak):
    """Sub-bin carrier offset from a parabola through |X[peak-1]|, |X[peak]|, |X[peak+1]|
    (reference carrier_interpolators.py:44-49)."""
    left, mid, right = fft_mag[peak - 1], fft_mag[peak], fft_mag[peak + 1]
    return (right - left) / (4 * mid - 2 * left - 2 * right)


def gaussian(fft_mag, peak):
    """The same on the logarithms of the three magnitudes (reference :52-58)."""
    left, mid, right = np.log(fft_mag[peak - 1]), np.log(fft_mag[peak]), np.log(fft_mag[peak + 1])
    return (right - left) / (4 * mid - 2 * left - 2 * right)


def cosine(fft_mag, peak):
    """Cosine fit through the three magnitudes (reference :92-100)."""
    left, mid, right = fft_mag[peak - 1], fft_mag[peak], fft_mag[peak + 1]
    cos_omega = (left + right) / (2 * mid)
    if cos_omega > 1:
        return 0
    omega = np.arccos(cos_omega)
    theta = np.arctan((left - right) / (2 * mid * np.sin(omega)))
    return -theta / omega


INTERPOLATORS = {"none": none, "parabolic": parabolic, "gaussian": gaussian, "cosine": cosine}
