"""Correlation peak interpolators for the experimental detector
(reference thrifty/experimental/xcorr_interpolators.py).

Any of them -- or any other callable `(corr_mag, peak_idx) -> offset` -- can be assigned to
`Detector.soa_estimate.interpolate`, as the reference's experiment does
(experimental/detect_xcorr_interpol.py:62).  The engine then keeps its verdicts and its peak search,
and the callable is EVALUATED here, on the host, on the correlation magnitudes of the detected
blocks (thrifty_amd.detect: the slow path); the result is clipped to +-0.6 sample like the
reference's SoaEstimator does (soa_estimator.py:16-17, :88).  The engine's own interpolator -- the
parabola through the logarithms, `gaussian` below -- needs none of this.
"""
import numpy as np


def _clamp(value, bound=0.5):
    """(reference xcorr_interpolators.py:27-28)"""
    return min(max(value, -bound), bound)


def _delay(samples, shift):
    """`samples` delayed by a fractional number of samples: a linear phase on its spectrum
    (reference :7-12)."""
    ramp = np.exp(-2j * np.pi * shift * np.fft.fftfreq(len(samples)))
    return np.fft.ifft(np.fft.fft(samples) * ramp)


def _xcorr_at(template, signal, lags):
    """sum_k signal[k + lag] conj(template[k]) over the overlap, for each of a few lags
    (reference :15-24)."""
    assert len(template) == len(signal)
    size = len(template)
    conj = np.conj(template)
    out = np.zeros(len(lags), dtype=signal.dtype)
    for j, lag in enumerate(lags):
        out[j] = np.sum(signal[max(0, lag):min(size, size + lag)] * conj[max(0, -lag):min(size, size - lag)])
    return out


def none(corr_mag, peak):
    """No sub-sample estimate (reference :31-32)."""
    return 0


def parabolic(corr_mag, peak):
    """Vertex of the parabola through the three magnitudes around the peak (reference :35-38)."""
    left, mid, right = corr_mag[peak - 1], corr_mag[peak], corr_mag[peak + 1]
    return 0.5 * (right - left) / (2 * mid - left - right)


def gaussian(corr_mag, peak):
    """The same through their logarithms (reference :41-45) -- what the engine computes itself
    (soa_estimator.py:165-171, k_finish)."""
    left, mid, right = np.log(corr_mag[peak - 1]), np.log(corr_mag[peak]), np.log(corr_mag[peak + 1])
    return 0.5 * (right - left) / (2 * mid - left - right)


def cosine(corr_mag, peak):
    """Cosine through the three magnitudes (reference :48-56); the int 0 where it has no solution."""
    left, mid, right = corr_mag[peak - 1], corr_mag[peak], corr_mag[peak + 1]
    cos_omega = (left + right) / (2 * mid)
    if cos_omega > 1:
        return 0
    omega = np.arccos(cos_omega)
    return -np.arctan((left - right) / (2 * mid * np.sin(omega))) / omega


def make_autocorr_fit(template):
    """Fit the measured peak, shifted by a sub-sample delay, to the template's own correlation
    against its on-off-keyed form (reference :59-92; SciPy's bounded curve_fit, as there)."""
    from scipy.optimize import curve_fit
    ook = (template - np.min(template)) * 2

    def autocorr_fit(corr_mag, peak, n=2):
        start = _clamp(gaussian(corr_mag, peak))
        lags = np.arange(-n, n + 1)
        measured = corr_mag[peak + lags]
        model = _xcorr_at(ook, template, lags)
        model *= np.sum(measured) / np.sum(model)

        def shifted(_lags, amplitude, offset):
            return amplitude * np.abs(_delay(measured, -offset))

        try:
            popt, _ = curve_fit(shifted, lags, model, p0=(1, start), bounds=([0.1, -0.55], [2, 0.55]),
                                sigma=np.abs(lags) + 1)
        except RuntimeError:            # "Optimal parameters not found": fall back to the three-point estimate
            return start
        return popt[1]

    return autocorr_fit


def make_maximise(template):
    """Maximise |sum_k X[k] conj(T[k]) e^{+2 pi i offset f_k}| over the sub-sample offset, on the
    template-long slice of the (carrier-synchronised) time-domain block at the peak
    (reference :95-112; SciPy's bounded `minimize`)."""
    from scipy.optimize import minimize
    template_spec = np.conj(np.fft.fft(template))

    def iterative(signal, peak, guess=0):
        cross = np.fft.fft(signal[peak:peak + len(template)]) * template_spec
        freqs = np.fft.fftfreq(len(cross))

        def cost(offset):
            return -np.abs(np.sum(cross * np.exp(2j * np.pi * offset * freqs)))

        return minimize(cost, guess, bounds=[(-0.55, 0.55)]).x[0]

    return iterative


# name -> interpolator, or (for the last two) a factory taking the template
INTERPOLATORS = {
    "none": none,
    "parabolic": parabolic,
    "gaussian": gaussian,
    "cosine": cosine,
    "autocorr": make_autocorr_fit,
    "maximise": make_maximise,
}
