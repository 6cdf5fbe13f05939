"""`Detector.sync` / `Detector.soa_estimate`: twins of the reference's `DefaultSynchronizer` and
`SoaEstimator` objects (carrier_sync.py:30-118, soa_estimator.py:63-102) -- the same attributes, and
callable like them, evaluated by the engine for one block; `interpolator` / `interpolate` can be
assigned a host callable (the slow paths of thrifty_amd.detect.Detector)."""
from __future__ import annotations

import numpy as np

from thrifty_amd import _native, toads_data


def unique_window(block_len, history_len, template_len):
    """Half-open range of correlation lags owned by one block (reference
    soa_estimator.py:20-39)."""
    assert history_len >= template_len - 1
    corr_len = block_len - template_len + 1
    pad = history_len - template_len + 1
    return pad // 2, corr_len - (pad - pad // 2)


class SyncStage(object):
    """`Detector.sync`: what the reference's `DefaultSynchronizer` offers an analysis script
    (carrier_sync.py:30-118) -- the attributes `thresh_coeffs`, `window`, `weights` and the call
    `sync(signal) -> (shifted_fft or None, CarrierSyncInfo)` -- evaluated by the engine for ONE
    block (carrier stage, Dirichlet fit, frequency shift, FFT#2; `thr_debug_stage` returns the
    shifted spectrum in natural order).  `detector` and `shifter` are stages of fused kernels and
    cannot be replaced (the reference's own subclasses that do so are separate detectors here:
    `PreshiftDetector`, `FastDetector`).  `interpolator` CAN be assigned, as the reference's
    InterpolationDetector does (experimental/detect_carrier_interpol.py:17-40): any callable
    `(fft_mag, peak_idx) -> offset` -- or None for no sub-bin estimate, carrier_sync.py:66-68 --
    then runs on the HOST between two engine passes (carrier stage + |FFT#1| out, offsets back in:
    thr_detect_offsets), a slow path for analysis scripts."""

    _DEVICE_FIT = object()      # `interpolator` not assigned: the engine's own Dirichlet fit (k_fit)

    def __init__(self, det, settings):
        self._det, self.weights = det, None
        self.thresh_coeffs, self.window = settings.carrier_thresh, settings.carrier_window
        # _last: (the shifted_fft handed out, record, corr) of the latest block
        self._last, self._interpolator = None, self._DEVICE_FIT

    @property
    def interpolator(self):
        return self._device_interpolator if self._interpolator is self._DEVICE_FIT else self._interpolator

    @interpolator.setter
    def interpolator(self, fn):
        if fn is not None and not callable(fn):
            raise TypeError("sync.interpolator takes a callable (fft_mag, peak_idx) -> offset, or None")
        self._interpolator = fn
        self._det._use_host_interpolator()

    def sync(self, signal):
        det = self._det
        if self._interpolator is not self._DEVICE_FIT:
            # (the stage dump is the engine's OWN pipeline, Dirichlet fit included: it cannot show
            # the spectrum shifted by somebody else's offset)
            raise NotImplementedError("sync(block) evaluates the engine's own stages; with a replaced "
                                      "interpolator use Detector.detect(timestamp, block_idx, block)")
        arr = det._stack([signal])
        rec = det._run(arr, np.zeros(1, dtype=np.int64))[0, 0]
        _, result = det._result(0.0, 0, rec)
        if result.corr_info is None:
            self._last = None
            return None, result.carrier_info
        xhat, corr = det._engine.debug_stage(arr)
        shifted_fft = xhat[0]
        self._last = (shifted_fft, rec, corr[0][:det.soa_estimate.corr_len])
        return shifted_fft, result.carrier_info

    __call__ = sync

    def detect(self, fft_mag):
        raise NotImplementedError(
            "the carrier detector runs inside the engine's carrier kernel, on a block's samples: "
            "call sync(block) -- or Detector.detect(timestamp, block_idx, block) -- instead")

    detector = detect

    def _device_interpolator(self, fft_mag, peak_idx):
        raise NotImplementedError("the Dirichlet fit runs inside the engine (k_fit): call sync(block) -- or "
                                  "assign sync.interpolator a host callable (slow path)")

    def shifter(self, signal, shift):
        raise NotImplementedError("the frequency shift is fused into the correlate kernel: call sync(block)")


class SoaStage(object):
    """`Detector.soa_estimate`: the attributes of the reference's `SoaEstimator`
    (soa_estimator.py:63-92: `template`, `template_energy`, `corr_len`, `window`,
    `thresh_coeffs`) and the call `soa_estimate(fft) -> (detected, CorrDetectionInfo, corr)` for
    the spectrum `Detector.sync(block)` has just returned -- the pair of calls that makes up the
    body of the reference's `Detector.detect` (detect.py:60-78).  Any other spectrum would have to be
    correlated from host memory, which the engine has no entry point for."""

    _DEVICE = object()          # `interpolate` not assigned: the engine's log-parabola (k_finish)

    def __init__(self, det, settings, template, corr_len):
        self._det = det
        self._interpolate = self._DEVICE
        self.last_fft = None        # the shifted spectrum of the block a replaced `interpolate` is looking at
        self.template = template
        self.template_energy = float(np.sum(np.abs(template) ** 2))
        self.corr_len = corr_len
        self.thresh_coeffs = settings.corr_thresh
        self.window = unique_window(settings.block_len, settings.history_len, template.shape[-1])

    @property
    def interpolate(self):
        """The correlation-peak interpolator (reference soa_estimator.py:74: `self.interpolate =
        gaussian_interpolation`).  Assignable like the reference's (experimental/
        detect_xcorr_interpol.py:62): any callable `(corr_mag, peak_idx) -> offset`, evaluated on the
        HOST for the detected blocks of a batch -- a slow path for analysis scripts."""
        return self._device_interpolate if self._interpolate is self._DEVICE else self._interpolate

    @interpolate.setter
    def interpolate(self, fn):
        if not callable(fn):
            raise TypeError("soa_estimate.interpolate takes a callable (corr_mag, peak_idx) -> offset")
        self._det._use_host_soa_interpolator()
        self._interpolate = fn

    def _device_interpolate(self, corr_mag, peak_idx):
        raise NotImplementedError("the log-parabola runs inside the engine (k_finish): call "
                                  "soa_estimate(shifted_fft) -- or assign soa_estimate.interpolate a host callable")

    def soa_estimate(self, fft):
        if self._interpolate is not self._DEVICE:
            raise NotImplementedError("soa_estimate(fft) evaluates the engine's own stages; with a replaced "
                                      "interpolator use Detector.detect(timestamp, block_idx, block)")
        last = self._det.sync._last
        if last is None or fft is not last[0]:
            raise NotImplementedError(
                "soa_estimate() takes the shifted spectrum that Detector.sync(block) returned for "
                "the latest block; arbitrary spectra cannot be handed to the engine")
        _, rec, corr = last
        detected = bool(int(rec["flags"]) & _native.FLAG_CORR)
        info = toads_data.CorrDetectionInfo(int(rec["corr_sample"]), float(rec["corr_offset"]) if detected else 0,
                                            float(rec["corr_energy"]), float(rec["corr_noise"]))
        return detected, info, corr

    __call__ = soa_estimate
