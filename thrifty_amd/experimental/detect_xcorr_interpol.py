#!/usr/bin/env python
"""Detector for experimenting with correlation peak interpolation methods (GPU counterpart of the
reference's thrifty/experimental/detect_xcorr_interpol.py:20-80).

Example usage:
    python -m thrifty_amd.experimental.detect_xcorr_interpol --method autocorr rx.card -o rx.toad

`InterpolationDetector(settings, blocks, rxid, method)` is the reference class: `method` names one of
`xcorr_interpolators.INTERPOLATORS` or is a callable `(corr_mag, peak_idx) -> offset`, assigned to
`self.soa_estimate.interpolate` exactly as the reference does.  The default `gaussian` IS the engine's
own interpolator and runs at full speed; every other choice is evaluated on the host on the
correlation of the detected blocks (a stage dump per batch) -- a slow path, as suits an experiment.
`maximise` is the reference's IterativeSoaEstimator (:20-34): it also looks at the carrier-synchronised
block itself, which `soa_estimate.last_fft` holds while the callable runs.
"""
from __future__ import print_function

import argparse

import numpy as np

from thrifty_amd.detect import Detector, detector_cli
from thrifty_amd.experimental import xcorr_interpolators


class InterpolationDetector(Detector):
    def __init__(self, settings, blocks=None, rxid=-1, method="gaussian", **kwargs):
        super(InterpolationDetector, self).__init__(settings, blocks, rxid, **kwargs)
        if callable(method):
            self.soa_estimate.interpolate = method
        elif method == "gaussian":
            pass                    # the engine's own log-parabola (soa_estimator.py:74, :165-171)
        elif method == "maximise":
            refine = xcorr_interpolators.make_maximise(settings.template)
            stage = self.soa_estimate

            def iterative_interpolate(corr_mag, peak_idx):
                # the time-domain block after carrier recovery (reference :31-34: `self._last_fft.ifft`)
                signal = np.fft.ifft(stage.last_fft)
                return refine(signal, peak_idx, xcorr_interpolators.gaussian(corr_mag, peak_idx))

            self.soa_estimate.interpolate = iterative_interpolate
        elif method == "autocorr":
            self.soa_estimate.interpolate = xcorr_interpolators.make_autocorr_fit(settings.template)
        elif method in ("none", "parabolic", "cosine"):
            self.soa_estimate.interpolate = xcorr_interpolators.INTERPOLATORS[method]
        else:
            raise KeyError("Unknown interpolation method")


def _main():
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--method", type=str, default="gaussian",
                        help="Correlation interpolation method. Valid methods are: "
                             + " ".join(sorted(xcorr_interpolators.INTERPOLATORS)))
    detector_cli(InterpolationDetector, parser, ["method"])


if __name__ == "__main__":
    _main()
