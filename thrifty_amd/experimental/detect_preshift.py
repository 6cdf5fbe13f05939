#!/usr/bin/env python
"""Detector that compensates the carrier offset with a bank of pre-shifted templates
instead of a second FFT (GPU counterpart of reference
thrifty/experimental/detect_preshift.py).

Example usage:
    python -m thrifty_amd.experimental.detect_preshift --num 101 rx.card

On the MI355X the variant collapses into ONE kernel per block at block_len 16384
(csrc/detect16k_preshift.hip): forward FFT, carrier verdict + parabolic interpolation,
roll-as-gather multiply with the nearest pre-shifted template spectrum, inverse FFT,
correlation statistics.
"""
from __future__ import print_function

import argparse

import numpy as np

from thrifty_amd.detect import Detector, detector_cli
from thrifty_amd.experimental.carrier_interpolators import parabolic

NUM_TEMPLATES = 21


class PreshiftDetector(Detector):
    """Same constructor as the reference (detect_preshift.py:48-60).  `interpolator` must be
    `parabolic` (the reference default and the only one with a device implementation);
    `corr_shift` is accepted and ignored -- the reference constructor forces it off too
    (detect_preshift.py:60)."""

    _fit_reach = 1                 # parabolic() reads fft_mag[peak + 1]
    _offset_type = np.float32      # float32 magnitudes in -> float32 offset out, as in the reference

    def __init__(self, settings, blocks=None, rxid=-1, yield_data=False, num=NUM_TEMPLATES,
                 interpolator=parabolic, corr_shift=False, batch_size=None, device_id=0):
        if interpolator is not parabolic and interpolator != "parabolic":
            raise NotImplementedError("only the parabolic carrier interpolator runs on the device")
        if yield_data:
            raise NotImplementedError("yield_data is not available in the preshift variant")
        if np.asarray(settings.template).ndim != 1:
            raise ValueError("PreshiftDetector takes one 1-D template")
        self.num = int(num)
        self.block_len = settings.block_len
        self.corr_shift = False
        super(PreshiftDetector, self).__init__(settings, blocks, rxid, yield_data,
                                               batch_size=batch_size, device_id=device_id,
                                               _preshift_num=self.num)
        self.shifts = np.linspace(-0.5, 0.5, self.num)   # TemplateShifts.shifts


def _main():
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--num", type=int, default=NUM_TEMPLATES,
                        help="Number of templates to precompute")
    detector_cli(PreshiftDetector, parser, ["num"])


if __name__ == "__main__":
    _main()
