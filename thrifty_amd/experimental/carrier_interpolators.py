"""Carrier peak interpolators selectable for the experimental detectors
(reference thrifty/experimental/carrier_interpolators.py).

On the GPU the interpolation happens inside the detection kernel; the functions here are
the *selectors* the reference API passes around, plus a host evaluation for analysis
scripts.  Only the ones with a device implementation are offered.
"""


def parabolic(fft_mag, peak):
    """Sub-bin carrier offset from a parabola through |X[peak-1]|, |X[peak]|, |X[peak+1]|
    (reference carrier_interpolators.py:44-49).  Device twin: preshift_verdict() in
    csrc/detect16k_preshift.hip."""
    left, mid, right = fft_mag[peak - 1], fft_mag[peak], fft_mag[peak + 1]
    return (right - left) / (4 * mid - 2 * left - 2 * right)


INTERPOLATORS = {"parabolic": parabolic}
