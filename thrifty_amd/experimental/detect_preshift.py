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
from thrifty_amd.experimental.carrier_interpolators import INTERPOLATORS, parabolic

NUM_TEMPLATES = 21


class PreshiftDetector(Detector):
    """Same constructor as the reference (detect_preshift.py:48-60).  `interpolator`: one of
    `carrier_interpolators.none / parabolic / gaussian / cosine` (the function or its name; the
    reference default is `parabolic`; `None` keeps the default too, as in the reference, where it
    leaves the Detector's own interpolator in place -- which here is not a device stage of this
    variant).  `corr_shift` is accepted and ignored -- the reference constructor forces it off too
    (detect_preshift.py:60)."""

    _fit_reach = 1                 # the three-point interpolators read fft_mag[peak + 1]
    _offset_type = np.float32      # float32 magnitudes in -> float32 offset out, as in the reference

    def __init__(self, settings, blocks=None, rxid=-1, yield_data=False, num=NUM_TEMPLATES,
                 interpolator=parabolic, corr_shift=False, batch_size=None, device_id=0):
        names = dict((fn, name) for name, fn in INTERPOLATORS.items())
        if interpolator is None:
            interpolator = parabolic
        name = names.get(interpolator, interpolator if isinstance(interpolator, str) else None)
        if name not in INTERPOLATORS:
            raise NotImplementedError("carrier interpolators with a device implementation: %s "
                                      "(pass the function from thrifty_amd.experimental."
                                      "carrier_interpolators or its name)" % ", ".join(sorted(INTERPOLATORS)))
        self.interpolator = name
        if name == "none":
            self._offset_type = int      # none() returns the int 0: the .toad column reads "0"
        if np.asarray(settings.template).ndim != 1:
            raise ValueError("PreshiftDetector takes one 1-D template")
        self.num = int(num)
        self.block_len = settings.block_len
        self.corr_shift = False
        super(PreshiftDetector, self).__init__(settings, blocks, rxid, yield_data,
                                               batch_size=batch_size, device_id=device_id,
                                               _preshift_num=self.num, _interpolator=name,
                                               # (stage dumps -- the rolled FFT#1 and the correlation the
                                               # reference returns under yield_data -- come from the
                                               # multi-pass kernels; the fused kernel keeps neither)
                                               _path="multipass" if yield_data else "auto")
        self.shifts = np.linspace(-0.5, 0.5, self.num)   # TemplateShifts.shifts


def _main():
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("--num", type=int, default=NUM_TEMPLATES,
                        help="Number of templates to precompute")
    detector_cli(PreshiftDetector, parser, ["num"])


if __name__ == "__main__":
    _main()
