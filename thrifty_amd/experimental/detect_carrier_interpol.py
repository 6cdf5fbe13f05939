#!/usr/bin/env python
"""Detector for experimenting with carrier peak interpolation methods (GPU counterpart of the
reference's thrifty/experimental/detect_carrier_interpol.py:17-59).

Example usage:
    python -m thrifty_amd.experimental.detect_carrier_interpol --method cosine rx.card -o rx.toad

`InterpolationDetector(settings, blocks, rxid, method, width)` is the reference class: `method` names
one of `carrier_interpolators.INTERPOLATORS` or is a callable `(fft_mag, peak_idx) -> offset`, which
is assigned to `self.sync.interpolator` exactly as the reference does.  The default Dirichlet fit
(`method='dirichlet'` with the reference's width of 6, i.e. seven points) IS the engine's own
interpolator and runs at full speed; every other choice runs on the host between two engine passes
(`thr_detect_offsets`) -- a slow path, as suits an experiment.
"""
from __future__ import print_function

import argparse

from thrifty_amd.detect import Detector, detector_cli
from thrifty_amd.experimental import carrier_interpolators


class InterpolationDetector(Detector):
    def __init__(self, settings, blocks=None, rxid=-1, method=None, width=6, **kwargs):
        super(InterpolationDetector, self).__init__(settings, blocks, rxid, **kwargs)
        if method is None:
            return
        if isinstance(method, str):
            if method == "dirichlet":
                if width == 6:
                    return          # the engine's own seven-point Dirichlet fit (carrier_sync.py:150-196)
                interpolator = carrier_interpolators.make_dirichlet(settings.block_len, settings.carrier_len, width)
            elif method in carrier_interpolators.INTERPOLATORS:
                interpolator = carrier_interpolators.INTERPOLATORS[method]
            else:
                raise KeyError("Unknown interpolation method")
            self.sync.interpolator = interpolator
        else:
            self.sync.interpolator = method


def _main():
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    names = sorted(list(carrier_interpolators.INTERPOLATORS) + ["dirichlet"])
    parser.add_argument("--method", type=str, default="dirichlet",
                        help="Carrier interpolation method. Valid methods are: " + " ".join(names))
    parser.add_argument("--width", type=int, default=6, help="Number of samples to use for interpolation")
    detector_cli(InterpolationDetector, parser, ["method", "width"])


if __name__ == "__main__":
    _main()
