"""NumPy restatement of the reference `thrifty detect` per-block algorithm.

TEST INFRASTRUCTURE ONLY.  This module is the parity oracle for the HIP path in
``thrifty_amd``; it is imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` and by nothing else.  The shipped product
never routes through it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
below against fixtures produced by importing the reference itself
(``tests/golden/make_golden.py``; NumPy 2.2.6 / SciPy 1.15.3, np.fft branch of
signal_utils.py:10-32) and against the known-answer tables of the reference's
own unit tests (tests/test_carrier_detect.py, test_carrier_sync.py,
test_soa_estimator.py, test_block_data.py, test_util.py).

Exception: the fastdet-compatible section near the end is PARITY UNPINNED (its header says
why); everything else is pinned as described above.

Beside it: ``oracle/_ref/libfastcard_readers.so`` (``oracle/Makefile``, ``oracle/ref_readers.py``)
-- the reference's own native block readers compiled from /root/reference -- checks the `.card` /
raw-stream front end a second time (tests/test_ref_readers.py).  The rest of fastcard / fastdet
needs FFTW3f / VOLK / librtlsdr / argp and cannot be built in this image.

Each function cites the reference file:line it follows (paths relative to the
reference checkout).  Third-party arithmetic: ``np.fft`` (pocketfft) and
``scipy.optimize.curve_fit`` (MINPACK lmdif) exactly as the reference calls
them (signal_utils.py:23-32, carrier_sync.py:189); the reference does not pin
their versions (requirements.txt:2-3).

dtype notes (NumPy >= 2 semantics, which the fixtures were generated under):
FFT #1 and the carrier statistics stay complex64/float32; the frequency shift,
FFT #2, the IFFT and all correlation statistics are complex128/float64
(carrier_sync.py:235-237 promotes).
"""
from __future__ import annotations

import math
from collections import namedtuple

import numpy as np
from scipy.optimize import curve_fit

CarrierStage = namedtuple("CarrierStage", "detected bin offset energy noise threshold")
CorrStage = namedtuple("CorrStage", "detected sample offset energy noise threshold")
BlockResult = namedtuple("BlockResult", "detected soa carrier corr")


# --------------------------------------------------------------------------
# a1: u8 IQ pairs -> complex64            (block_data.py:38-52, rawconv.c:10-26)
# --------------------------------------------------------------------------
def iq_u8_to_c64(raw):
    raw = np.asarray(raw, dtype=np.uint8)
    z = raw.astype(np.float32).view(np.complex64)
    z -= 127.4 + 127.4j
    z /= 128
    return z


def c64_to_iq_u8(z):
    """Inverse quantiser (block_data.py:55-67): *128 + 127.4, truncate."""
    f = np.asarray(z).astype(np.complex64).view(np.float32) * 128 + 127.4
    return f.astype(np.uint8)


# --------------------------------------------------------------------------
# a5: carrier window + threshold detector          (carrier_detect.py:17-154)
# --------------------------------------------------------------------------
def window_to_indices(start, stop, n):
    """Closed bin interval -> FFT index interval (carrier_detect.py:17-58).

    ``stop`` may come back >= n, meaning the window wraps."""
    if abs(start) >= n or abs(stop) >= n:
        raise ValueError("Frequency window out of range: {} - {}".format(start, stop))
    if start < 0 <= stop:
        start, stop = n + start, n + stop
    if start < 0:
        start += n
    if stop < 0:
        stop += n
    if stop < start:
        start, stop = stop, start
    return start, stop


def carrier_peak(mag, window):
    """First-max inside the (wrapping, inclusive) window (carrier_detect.py:118-154
    with peak_filter=None).  Keeps the reference's ``> len`` quirk."""
    n = len(mag)
    lo, hi = window_to_indices(*(window if window is not None else (0, -1)), n)
    sel = np.take(mag, range(lo, hi + 1), mode="wrap")
    rel = int(np.argmax(sel))
    peak_mag = sel[rel]
    idx = rel + lo
    if idx > n:
        idx -= n
    return idx, peak_mag


def carrier_noise(mag, peak_mag):
    """sqrt((sum mag^2 - 2 peak^2)/(N-1))  (carrier_detect.py:99-107)."""
    total = np.sum(mag ** 2)
    with np.errstate(invalid="ignore"):
        return np.sqrt((total - 2 * peak_mag ** 2) / (len(mag) - 1))


def threshold_value(mag, coeffs, noise_rms):
    """sqrt(c + s*noise^2 + d*std(mag)^2)  (carrier_detect.py:110-115,
    soa_estimator.py:127-134)."""
    c, s, d = coeffs
    std = np.std(mag) if d else 0
    with np.errstate(invalid="ignore"):
        return np.sqrt(c + s * noise_rms ** 2 + d * std ** 2)


def carrier_detect(mag, coeffs, window):
    """(detected, bin, peak_mag, noise_rms, threshold)  (carrier_detect.py:61-96)."""
    idx, peak = carrier_peak(mag, window)
    noise = carrier_noise(mag, peak)
    thr = threshold_value(mag, coeffs, noise)
    return bool(peak > thr), idx, peak, noise, thr


# --------------------------------------------------------------------------
# a6: Dirichlet-kernel sub-bin fit                  (carrier_sync.py:121-196)
# --------------------------------------------------------------------------
def dirichlet(x, n, w):
    """sin(pi W x/N) / (W sin(pi x/N)), 1 at x=0  (carrier_sync.py:121-132)."""
    x = np.array(x, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = np.sin(np.pi * w * x / n) / np.sin(np.pi * x / n) / w
        d[np.isnan(d)] = 1
    return d


def dirichlet_fit(mag, peak_idx, n, w, width=6):
    """Least-squares (A, offset) of A*|D(x-offset)| over peak_idx + [-3..3],
    SciPy curve_fit from p0 = (mag[peak], 0)  (carrier_sync.py:179-194).
    Indexing is NumPy's: negative indices wrap, >= N raises IndexError."""
    xs = np.arange(-(width // 2), width // 2 + 1)
    ys = mag[peak_idx + xs]

    def model(x, amp, off):
        return amp * np.abs(dirichlet(np.array(x, dtype=np.float64) - off, n, w))

    popt, _ = curve_fit(model, xs, ys, p0=(mag[peak_idx], 0))
    return popt[0], popt[1]


# --------------------------------------------------------------------------
# a7: fractional frequency shift + FFT #2           (carrier_sync.py:222-238)
# --------------------------------------------------------------------------
def shift_and_fft(x, shift):
    n = len(x)
    ramp = np.arange(n) * 1.0 / n - 0.5
    rot = np.exp(2j * np.pi * shift * ramp)
    return np.fft.fft(x * rot)


# --------------------------------------------------------------------------
# a9-a14: matched filter + SoA estimator               (soa_estimator.py)
# --------------------------------------------------------------------------
def unique_window(block_len, history_len, template_len):
    """Half-open range of lags unique to a block (soa_estimator.py:20-39)."""
    assert history_len >= template_len - 1
    corr_len = block_len - template_len + 1
    pad = history_len - template_len + 1
    left = pad // 2
    return left, corr_len - (pad - left)


class TemplateBank(object):
    """Per-template constants (soa_estimator.py:63-76)."""

    def __init__(self, template, block_len, history_len):
        template = np.asarray(template)
        self.template = template
        self.energy = np.sum(np.abs(template) ** 2)
        self.template_len = len(template)
        self.corr_len = block_len - self.template_len + 1
        padded = np.concatenate([template, np.zeros(self.corr_len - 1)])
        self.spectrum = np.fft.fft(padded)
        # cached like signal_utils.py:130-136: a *named* conjugate keeps NumPy from
        # eliding a temporary into an in-place multiply (different rounding path)
        self.spectrum_conj = np.conj(self.spectrum)
        self.window = unique_window(block_len, history_len, self.template_len)


def despread(xhat, bank):
    """ifft(X * conj(T))[:corr_len]  (soa_estimator.py:97-102)."""
    prod = xhat * bank.spectrum_conj
    return np.fft.ifft(prod)[: bank.corr_len]


def corr_peak(corr_mag, window):
    lo, hi = window
    idx = int(np.argmax(corr_mag[lo:hi])) + lo
    return idx, corr_mag[idx]


def corr_noise(xhat, bank, peak_mag):
    """sqrt((mean|X|^2 * sum t^2 - peak^2)/N)  (soa_estimator.py:108-120,
    signal_utils.py:118-128)."""
    energy = np.sqrt(np.mean(np.abs(xhat) ** 2)) ** 2
    with np.errstate(invalid="ignore"):
        return np.sqrt((energy * bank.energy - peak_mag ** 2) / len(xhat))


def log_parabola(corr_mag, idx):
    """Gaussian 3-point interpolation (soa_estimator.py:159-170)."""
    if idx == 0 or idx == len(corr_mag) - 1:
        return 0
    a, b, c = np.log(corr_mag[idx - 1]), np.log(corr_mag[idx]), np.log(corr_mag[idx + 1])
    return 0.5 * (c - a) / (2 * b - a - c)


def clip_offset(v, lim=0.6):
    return -lim if v < -lim else lim if v > lim else v


def soa_estimate(xhat, bank, coeffs, interpolate=None):
    """(CorrStage, corr)  (soa_estimator.py:78-92).  interpolate: None = the SoaEstimator's own
    gaussian_interpolation, or what the reference's experiment assigns to `soa_estimate.interpolate`
    (experimental/detect_xcorr_interpol.py:36-62), here called as (corr_mag, peak_idx, xhat)."""
    corr = despread(xhat, bank)
    mag = np.abs(corr)
    idx, peak = corr_peak(mag, bank.window)
    noise = corr_noise(xhat, bank, peak)
    thr = threshold_value(mag, coeffs, noise)
    det = bool(peak > thr)
    if interpolate is None:
        off = clip_offset(log_parabola(mag, idx) if det else 0)
    else:
        off = clip_offset(interpolate(mag, idx, xhat) if det else 0)
    return CorrStage(det, idx, off, peak, noise, thr), corr


# -- what the experiment offers for `interpolate`      (experimental/xcorr_interpolators.py)
def xcorr_none(mag, idx, xhat=None):
    """xcorr_interpolators.py:31-32."""
    return 0


def xcorr_parabolic(mag, idx, xhat=None):
    """xcorr_interpolators.py:35-38: no edge rule, the three magnitudes as they are."""
    a, b, c = mag[idx - 1], mag[idx], mag[idx + 1]
    return 0.5 * (c - a) / (2 * b - a - c)


def xcorr_gaussian(mag, idx, xhat=None):
    """xcorr_interpolators.py:41-45."""
    a, b, c = np.log(mag[idx - 1]), np.log(mag[idx]), np.log(mag[idx + 1])
    return 0.5 * (c - a) / (2 * b - a - c)


def xcorr_cosine(mag, idx, xhat=None):
    """xcorr_interpolators.py:48-56."""
    a, b, c = mag[idx - 1], mag[idx], mag[idx + 1]
    cw = (a + c) / (2 * b)
    if cw > 1:
        return 0
    w = np.arccos(cw)
    return -np.arctan((a - c) / (2 * b * np.sin(w))) / w


def xcorr_autocorr(template):
    """xcorr_interpolators.py:59-92: the five magnitudes around the peak, delayed by `offset` through
    their own spectrum, least-squares against the template's correlation with its on-off-keyed form
    at the same five lags (sigma |lag| + 1, amplitude in [0.1, 2], offset in +-0.55)."""
    import scipy.optimize
    template = np.asarray(template)
    ook = (template - template.min()) * 2
    size = len(template)

    def fit(mag, idx, xhat=None, n=2):
        guess = clip_offset(xcorr_gaussian(mag, idx), 0.5)
        lags = np.arange(-n, n + 1)
        seen = mag[idx + lags]
        want = np.zeros(len(lags), dtype=template.dtype)
        for j, lag in enumerate(lags):
            want[j] = np.sum(template[max(0, lag):min(size, size + lag)]
                             * np.conj(ook)[max(0, -lag):min(size, size - lag)])
        want *= np.sum(seen) / np.sum(want)
        freqs = np.fft.fftfreq(len(seen))

        def model(_x, amplitude, offset):
            return amplitude * np.abs(np.fft.ifft(np.fft.fft(seen) * np.exp(2j * np.pi * offset * freqs)))

        try:
            popt, _ = scipy.optimize.curve_fit(model, lags, want, p0=(1, guess),
                                               bounds=([0.1, -0.55], [2, 0.55]), sigma=np.abs(lags) + 1)
        except RuntimeError:
            return guess
        return popt[1]

    return fit


def xcorr_maximise(template):
    """IterativeSoaEstimator (detect_xcorr_interpol.py:20-34) with xcorr_interpolators.py:95-112: on
    the template-long slice of ifft(xhat) at the peak, maximise the correlation with the template
    delayed by a sub-sample offset in +-0.55, starting from the log-parabola's value."""
    import scipy.optimize
    spec = np.conj(np.fft.fft(template))

    def refine(mag, idx, xhat):
        signal = np.fft.ifft(xhat)
        cross = np.fft.fft(signal[idx:idx + len(template)]) * spec
        freqs = np.fft.fftfreq(len(cross))
        res = scipy.optimize.minimize(lambda o: -np.abs(np.sum(cross * np.exp(2j * np.pi * o * freqs))),
                                      xcorr_gaussian(mag, idx), bounds=[(-0.55, 0.55)])
        return res.x[0]

    return refine


# --------------------------------------------------------------------------
# a8 + a15: one block end to end        (carrier_sync.py:52-76, detect.py:60-78)
# --------------------------------------------------------------------------
class OracleDetector(object):
    """Functional twin of reference ``Detector`` for ONE template or several.

    ``detect_block`` returns one BlockResult per template (the reference has a
    single template; with several the carrier stage is shared, which is what
    running the reference once per template yields)."""

    def __init__(self, block_len, history_len, templates, carrier_thresh,
                 carrier_window, corr_thresh, carrier_len=None, interpolator="dirichlet", interpolate=None):
        """interpolator: "dirichlet" (the default Synchronizer's, carrier_sync.py:150-196), None (no
        sub-bin estimate, carrier_sync.py:66-68) or a callable (mag, peak_idx) -> offset -- what the
        reference's InterpolationDetector assigns to `sync.interpolator`
        (experimental/detect_carrier_interpol.py:17-40)."""
        self.interpolator = interpolator
        self.interpolate = interpolate      # of the correlation peak: see soa_estimate()
        if isinstance(templates, np.ndarray) and templates.ndim == 1:
            templates = [templates]
        self.block_len = block_len
        self.history_len = history_len
        self.new_len = block_len - history_len
        self.banks = [TemplateBank(t, block_len, history_len) for t in templates]
        self.carrier_len = carrier_len if carrier_len is not None else len(templates[0])
        self.carrier_thresh = carrier_thresh
        self.carrier_window = carrier_window
        self.corr_thresh = corr_thresh

    def carrier_stage(self, x):
        spec = np.fft.fft(x)
        mag = np.abs(spec)
        det, idx, peak, noise, thr = carrier_detect(mag, self.carrier_thresh,
                                                    self.carrier_window)
        off = 0
        xhat = None
        if det:
            if self.interpolator == "dirichlet":
                _, off = dirichlet_fit(mag, idx, self.block_len, self.carrier_len)
            elif self.interpolator is not None:
                off = self.interpolator(mag, idx)
            # (the reference's peak_idx is a NumPy int64 -- np.argmax + start, carrier_detect.py:138-154 --
            # so a float32 offset is widened: int64 + float32 -> float64, carrier_sync.py:71)
            xhat = shift_and_fft(x, -(np.int64(idx) + off))
        return CarrierStage(det, idx, off, peak, noise, thr), xhat, mag

    def detect_block(self, block_idx, x, want_data=False):
        assert len(x) == self.block_len
        car, xhat, _ = self.carrier_stage(x)
        out, data = [], []
        for bank in self.banks:
            if xhat is None:
                out.append(BlockResult(False, None, car, None))
                data.append((None, None))
                continue
            cs, corr = soa_estimate(xhat, bank, self.corr_thresh, self.interpolate)
            soa = self.new_len * block_idx + cs.sample + cs.offset
            out.append(BlockResult(cs.detected, soa, car, cs))
            data.append((xhat, corr))
        return (out, data) if want_data else out

    def detect_u8(self, block_idx, raw, want_data=False):
        return self.detect_block(block_idx, iq_u8_to_c64(raw), want_data)


# --------------------------------------------------------------------------
# 8(f) rank 2: PreshiftDetector              (experimental/detect_preshift.py)
# --------------------------------------------------------------------------
def parabolic_offset(mag, peak_idx):
    """3-point parabola on the FFT magnitudes (experimental/carrier_interpolators.py:40-45).
    `mag` is float32, so is the result; mag[peak-1] wraps for peak 0 (Python negative
    index) and mag[peak+1] raises IndexError past the end, like the reference."""
    a, b, c = mag[peak_idx - 1], mag[peak_idx], mag[peak_idx + 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        return (c - a) / (4 * b - 2 * a - 2 * c)


def no_offset(mag, peak_idx):
    """experimental/carrier_interpolators.py:17-18."""
    return 0


def gaussian_offset(mag, peak_idx):
    """The parabola through the LOGARITHMS of the three magnitudes
    (experimental/carrier_interpolators.py:48-54; float32 in, float32 out)."""
    a, b, c = mag[peak_idx - 1], mag[peak_idx], mag[peak_idx + 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        a, b, c = np.log(a), np.log(b), np.log(c)
        return (c - a) / (4 * b - 2 * a - 2 * c)


def cosine_offset(mag, peak_idx):
    """experimental/carrier_interpolators.py:84-92."""
    a, b, c = mag[peak_idx - 1], mag[peak_idx], mag[peak_idx + 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        cos_omega = (a + c) / (2 * b)
        if cos_omega > 1:
            return 0
        omega = np.arccos(cos_omega)
        theta = np.arctan((a - c) / (2 * b * np.sin(omega)))
        return -theta / omega


CARRIER_INTERPOLATORS = {"parabolic": parabolic_offset, "none": no_offset,
                         "gaussian": gaussian_offset, "cosine": cosine_offset}


def parabole_fit_offset(width):
    """experimental/carrier_interpolators.py:57-66: vertex of the least-squares parabola through
    the width + 1 magnitudes around the peak."""
    def interpolate(mag, peak_idx):
        x = np.arange(-(width // 2), width // 2 + 1)
        coeffs = np.polyfit(x, mag[peak_idx + x], 2)
        return -coeffs[1] / coeffs[0] / 2
    return interpolate


def corr_parabolic_offset(corr_width, block_len, carrier_len):
    """experimental/carrier_interpolators.py:69-81: the three-point parabola on the magnitudes
    correlated with the Dirichlet kernel (its own kernel: no absolute value, 1 at x = 0, :7-14)."""
    rel = np.arange(-(corr_width // 2), corr_width // 2 + 1)
    xr = np.array(rel, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        kern = np.sin(np.pi * carrier_len * xr / block_len) / np.sin(np.pi * xr / block_len) / carrier_len
        kern[np.isnan(kern)] = 1

    def interpolate(mag, peak_idx):
        a = np.sum(mag[peak_idx + rel - 1] * kern)
        b = np.sum(mag[peak_idx + rel] * kern)
        c = np.sum(mag[peak_idx + rel + 1] * kern)
        return (c - a) / (4 * b - 2 * a - 2 * c)
    return interpolate


class PreshiftBank(object):
    """`num` template spectra, pre-shifted by -0.5 .. 0.5 bins (detect_preshift.py:24-45)."""

    def __init__(self, template, block_len, num=21):
        template = np.asarray(template)
        self.corr_len = block_len - len(template) + 1
        padded = np.concatenate([template, np.zeros(self.corr_len - 1)])
        self.shifts = np.linspace(-0.5, 0.5, num)
        ramp = np.arange(block_len) * 1.0 / block_len - 0.5
        self.spectra_conj = []
        for shift in self.shifts:
            rot = np.exp(-2j * np.pi * shift * ramp)
            self.spectra_conj.append(np.conj(np.fft.fft(padded * rot)))
        self.num = num

    def nearest(self, frac):
        assert -0.5 <= frac <= 0.5
        return int(np.round((frac + 0.5) * (self.num - 1)))


class OraclePreshiftDetector(object):
    """Functional twin of the reference ``PreshiftDetector`` (defaults: num=21, parabolic
    interpolator, corr_shift off -- the constructor forces it off, detect_preshift.py:60):
    FFT#1 is rolled by the rounded carrier bin instead of re-transforming the shifted
    block, and the residual sub-bin offset picks the nearest pre-shifted template."""

    def __init__(self, block_len, history_len, template, carrier_thresh, carrier_window,
                 corr_thresh, num=21, interpolator="parabolic"):
        self.block_len, self.history_len = block_len, history_len
        self.new_len = block_len - history_len
        self.bank = TemplateBank(template, block_len, history_len)   # energy, window
        self.shifted = PreshiftBank(template, block_len, num)
        self.interpolate = CARRIER_INTERPOLATORS[interpolator]   # detect_preshift.py:57-58
        self.carrier_thresh, self.carrier_window = carrier_thresh, carrier_window
        self.corr_thresh = corr_thresh
        self.last = None        # (int_shift, frac_shift, template index) of the last block

    def detect_block(self, block_idx, x, want_data=False):
        """want_data: also return (rolled FFT#1, correlation) -- what the reference class hands out
        under yield_data (detect.py:60-78 with the shifter / despreader of detect_preshift.py:62-80)."""
        out = self._detect_block(block_idx, x)
        data, self._data = self._data, None
        return (out, data) if want_data else out

    def _detect_block(self, block_idx, x):
        assert len(x) == self.block_len
        self._data = None
        spec = np.fft.fft(x)                       # complex64 in -> complex64 out
        mag = np.abs(spec)
        det, idx, peak, noise, thr = carrier_detect(mag, self.carrier_thresh,
                                                    self.carrier_window)
        off = 0
        self.last = None
        if not det:
            return BlockResult(False, None, CarrierStage(det, idx, off, peak, noise, thr), None)
        off = self.interpolate(mag, idx)           # carrier_sync.py:68-70
        car = CarrierStage(det, idx, off, peak, noise, thr)
        # the reference's peak index is an np.int64 (argmax + start), so int64 + float32 -> float64
        shift = -(np.int64(idx) + off)             # carrier_sync.py:71
        int_shift = int(np.round(shift))           # detect_preshift.py:62-65
        frac = shift - int_shift
        rolled = np.roll(spec, int_shift)          # carrier_sync.py:241-245
        j = self.shifted.nearest(frac)
        self.last = (int_shift, frac, j)
        corr = np.fft.ifft(rolled * self.shifted.spectra_conj[j])[: self.shifted.corr_len]
        self._data = (rolled, corr)
        # the rest is soa_estimator.py:78-92 unchanged (noise from the rolled complex64 FFT#1)
        cmag = np.abs(corr)
        pk, peak_mag = corr_peak(cmag, self.bank.window)
        cnoise = corr_noise(rolled, self.bank, peak_mag)
        cthr = threshold_value(cmag, self.corr_thresh, cnoise)
        cdet = bool(peak_mag > cthr)
        coff = clip_offset(log_parabola(cmag, pk) if cdet else 0)
        cs = CorrStage(cdet, pk, coff, peak_mag, cnoise, cthr)
        return BlockResult(cdet, self.new_len * block_idx + pk + coff, car, cs)

    def detect_u8(self, block_idx, raw, want_data=False):
        return self.detect_block(block_idx, iq_u8_to_c64(raw), want_data)


# --------------------------------------------------------------------------
# 8(f) rank 3: identify -- TX classification + duplicate filter   (identify.py)
# --------------------------------------------------------------------------
def transmitter_windows(freqs):
    """Window edges from the histogram of carrier bins (identify.py:26-77): a window opens
    where the count exceeds 1.25 std and closes where it drops below 0.4 std; edges are the
    mid-points between consecutive windows, framed by the first bin and last bin + 1."""
    freqs = np.asarray(freqs)
    first_bin = np.min(freqs)
    cnts = np.bincount(freqs - first_bin)
    last_bin = first_bin + len(cnts)
    low, high = np.std(cnts) * 0.4, np.std(cnts) * 1.25
    peaks, below, start = [], True, None
    for i, cnt in enumerate(cnts):
        if not below and cnt < low:
            peaks.append((start, i))
            start, below = None, True
        if below and cnt > high:
            start, below = i, False
    if not below:
        peaks.append((start, len(cnts) - 1))
    mids = [(peaks[i][1] + peaks[i + 1][0]) // 2 for i in range(len(peaks) - 1)]
    return np.concatenate([[first_bin], np.array(mids) + first_bin, [last_bin]])


def auto_classify(rxid, carrier_bin):
    """txid per detection = index of its RX's window (identify.py:80-106)."""
    rxid, carrier_bin = np.asarray(rxid), np.asarray(carrier_bin)
    txid = np.zeros(len(rxid), dtype=np.int64)
    edges = {}
    for rx in np.unique(rxid):
        sel = rxid == rx
        edges[int(rx)] = transmitter_windows(carrier_bin[sel])
        txid[sel] = np.digitize(carrier_bin[sel], edges[int(rx)][:-1]) - 1
    return txid, edges


def classify_by_map(rxid, carrier_bin, carrier_offset, freqmap):
    """freqmap: {rxid: {txid: (lo, hi)}} in insertion order; inclusive ranges on
    bin + offset, the LAST matching range wins, -1 if none (identify.py:109-121)."""
    txid = np.full(len(rxid), -1, dtype=np.int64)
    for i in range(len(rxid)):
        freq = carrier_bin[i] + carrier_offset[i]
        for tx, (lo, hi) in freqmap[int(rxid[i])].items():
            if lo <= freq <= hi:
                txid[i] = tx
    return txid


def duplicate_mask(rxid, txid, block, timestamp, energy):
    """True = keep (identify.py:140-172): sort by (rxid, txid, block, timestamp); drop a
    detection whose sorted neighbour sits in the adjacent block with more energy (the
    neighbour test does not look at rxid/txid and wraps around the ends, like np.roll), and
    every unidentified (-1) one."""
    order = np.lexsort((timestamp, block, txid, rxid))
    blk, en, tx = np.asarray(block)[order], np.asarray(energy)[order], np.asarray(txid)[order]
    pb, pe = np.roll(blk, 1), np.roll(en, 1)
    nb, ne = np.roll(blk, -1), np.roll(en, -1)
    drop = ((blk == pb + 1) & (en < pe)) | ((blk == nb - 1) & (en < ne)) | (tx == -1)
    mask = np.empty(len(order), dtype=bool)
    mask[order] = ~drop
    return mask


def filter_order(mask, timestamp):
    """Indices of the kept detections in output order: by timestamp, stable
    (identify.py:175-181)."""
    kept = np.flatnonzero(mask)
    return kept[np.argsort(np.asarray(timestamp)[kept], kind="stable")]


# --------------------------------------------------------------------------
# 8(f) rank 4: fastdet-compatible mode   (fastcard/cardet.c, fastdet/corr_detector.cpp)
#
# PARITY UNPINNED: fastdet needs FFTW3f, VOLK, librtlsdr and argp and cannot be built in this
# environment, and the reference has no fixtures for it.  This is a restatement of the C/C++
# sources (float32 where they use float); summation orders of VOLK's accumulators are not
# reproduced.  It pins the GPU variant to the restatement, not to fastdet's own output.
# --------------------------------------------------------------------------
FastdetResult = namedtuple("FastdetResult",
                           "carrier detected argmax carrier_max carrier_noise carrier_offset "
                           "peak_idx peak_offset peak_power noise_power soa")


def fastdet_window(start, stop, n):
    """cardet_normalize_window (cardet.c:43-70): inclusive [min, max], no wrap-around."""
    if start < 0 <= stop:
        raise ValueError("Carrier frequency window range not supported.")
    if start < 0:
        start += n
    if stop < 0:
        stop += n
    if start >= n or stop >= n:
        raise ValueError("Carrier frequency window out of range.")
    return (stop, start) if stop < start else (start, stop)


class OracleFastdet(object):
    """cardet_detect (cardet.c:7-41) + CorrDetector::detect (corr_detector.cpp:125-197)."""

    def __init__(self, block_len, history_len, template, carrier_thresh, carrier_window,
                 corr_thresh):
        self.n, self.new_len = block_len, block_len - history_len
        template = np.asarray(template, dtype=np.float32)
        self.tenergy = np.float32(np.sum(template.astype(np.float32) ** 2, dtype=np.float32))
        padded = np.zeros(block_len, dtype=np.complex64)
        padded[:len(template)] = template
        self.tconj = np.conj(np.fft.fft(padded)).astype(np.complex64)
        self.corr_len = block_len - len(template) + 1
        self.window = unique_window(block_len, history_len, len(template))   # set_window, 72-86
        self.cwin = fastdet_window(carrier_window[0], carrier_window[1], block_len)
        self.cthr = (np.float32(carrier_thresh[0]), np.float32(carrier_thresh[1]))
        self.xthr = (np.float32(corr_thresh[0]), np.float32(corr_thresh[1]))

    @staticmethod
    def _clip(v):
        return -0.5 if v < -0.5 else 0.5 if v > 0.5 else v

    def detect_block(self, block_idx, x):
        f32 = np.float32
        spec = np.fft.fft(np.asarray(x, dtype=np.complex64)).astype(np.complex64)
        power = (spec.real.astype(f32) ** 2 + spec.imag.astype(f32) ** 2).astype(f32)
        total = f32(np.sum(power, dtype=np.float32))
        lo, hi = self.cwin
        argmax = int(np.argmax(power[lo:hi + 1])) + lo
        mx = power[argmax]
        noise = f32(0) if total == 0 else f32((total - f32(2) * mx) / f32(self.n - 1))
        thr = f32(self.cthr[0] + self.cthr[1] * noise)
        if not mx > thr:
            return FastdetResult(False, False, argmax, mx, noise, 0.0, -1, 0.0, 0.0, 0.0, None)
        a, b, c = (math.sqrt(float(power[(argmax + d) % self.n])) for d in (-1, 0, 1))
        coff = self._clip((c - a) / (4 * b - 2 * a - 2 * c))              # interpolate_parabolic
        rolled = np.roll(spec, -argmax)
        corr = (np.fft.ifft(rolled * self.tconj)[:self.corr_len]).astype(np.complex64)
        cpow = (corr.real.astype(f32) ** 2 + corr.imag.astype(f32) ** 2).astype(f32)
        w0, w1 = self.window
        pk = int(np.argmax(cpow[w0:w1])) + w0
        peak = cpow[pk]
        signal_energy = f32(total / f32(self.n))
        # estimate_noise(size_t peak_power, ...): the peak power arrives truncated to an integer
        noise_power = f32((f32(signal_energy * self.tenergy) - f32(int(peak))) / f32(self.n))
        if noise_power < 0:
            noise_power = f32(0)
        det = bool(peak > f32(self.xthr[0] + self.xthr[1] * noise_power))
        off = 0.0
        if det and 0 < pk < self.corr_len - 1:
            la, lb, lc = (math.log(math.sqrt(float(cpow[pk + d]))) for d in (-1, 0, 1))
            off = self._clip((lc - la) / (4 * lb - 2 * la - 2 * lc))       # interpolate_gaussian
        soa = self.new_len * block_idx + pk + off                           # fastdet.cpp:184-185
        return FastdetResult(True, det, argmax, mx, noise, coff, pk, off, peak, noise_power, soa)

    def detect_u8(self, block_idx, raw):
        return self.detect_block(block_idx, iq_u8_to_c64(raw))


def fastdet_toad_line(rxid, timestamp, block_idx, r):
    """The line fastdet prints (fastdet.cpp:188-206): fixed precisions, roots of the powers."""
    sec = int(timestamp)
    usec = int(round((timestamp - sec) * 1e6))
    return "%d %d.%06d %d %.8f %u %.12f %f %f %u %f %f %f" % (
        rxid, sec, usec, block_idx, r.soa, r.peak_idx, r.peak_offset, math.sqrt(r.peak_power),
        math.sqrt(r.noise_power), r.argmax, r.carrier_offset, math.sqrt(r.carrier_max),
        math.sqrt(r.carrier_noise))


# --------------------------------------------------------------------------
# a16: .toad line                                      (toads_data.py:47-61)
# --------------------------------------------------------------------------
def toad_line(rxid, timestamp, block_idx, res):
    car, cs = res.carrier, res.corr
    s = ("{t:.6f} {b} {s:.8f} {ps} {po} {pe} {pn} {cb} {co} {ce} {cn}".format(
        t=timestamp, b=block_idx, s=res.soa, ps=cs.sample, po=cs.offset,
        pe=cs.energy, pn=cs.noise, cb=car.bin, co=car.offset, ce=car.energy,
        cn=car.noise))
    if rxid is not None:
        s = str(rxid) + " " + s
    return s


def fft_bin(idx, n):
    """FFT index -> signed bin (util.py:11-22)."""
    return idx if (idx < 0 or idx <= (2 * n - 1) / 4) else idx - n
