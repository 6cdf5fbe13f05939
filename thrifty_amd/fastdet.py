#!/usr/bin/env python
"""fastdet-compatible detector on the GPU.

The reference's native `fastdet` (fastcard/cardet.c + fastdet/corr_detector.cpp) is not the
Python `Detector` in C: it thresholds in the power domain, rolls FFT#1 by the integer carrier
bin instead of re-transforming a fractionally shifted block, interpolates with a parabola /
Gaussian clipped to +-0.5 and prints fixed-precision .toad lines.  `FastDetector` runs that
algorithm through `thr_create_fastdet` (one fused kernel per block at block_len 16384) behind
the same iteration protocol as `thrifty_amd.detect.Detector`.

    python -m thrifty_amd.fastdet rx.card -o rx.toad --tpl template.tpl [-c detector.cfg]

Thresholds given as `<const>c + <k>*snr` are interpreted in the POWER domain, like fastdet's
`--carrier-threshold` / `--corr-threshold`.
"""
from __future__ import print_function

import argparse
import math
import struct
import sys

import numpy as np

from thrifty_amd import _native, toads_data
from thrifty_amd.block_data import CardStream, RawStream
from thrifty_amd.detect import Detector, DetectorSettings
from thrifty_amd.settings import load_args
from thrifty_amd.setting_parsers import normalize_freq_range


def load_tpl(path):
    """.tpl = native-endian int16 length + float32[length]
    (reference scripts/npy_to_tpl.py:20-22, corr_detector.cpp:212-219)."""
    with open(path, "rb") as f:
        (length,) = struct.unpack("=h", f.read(2))
        data = np.frombuffer(f.read(4 * length), dtype="=f4")
    if len(data) != length:
        raise ValueError("%s: truncated template (%d of %d samples)" % (path, len(data), length))
    return data.astype(np.float32)


def save_tpl(path, template):
    template = np.asarray(template, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(struct.pack("=h", len(template)))
        f.write(template.astype("=f4").tobytes())


def fastdet_line(result):
    """One .toad line the way fastdet prints it (fastdet.cpp:188-206)."""
    cor, car = result.corr_info, result.carrier_info
    sec, usec = divmod(int(round(result.timestamp * 1e6)), 1000000)   # (carries into the seconds)
    return "%d %d.%06d %d %.8f %u %.12f %f %f %u %f %f %f" % (
        result.rxid, sec, usec, result.block, result.soa, cor.sample, cor.offset, cor.energy,
        cor.noise, car.bin, car.offset, car.energy, car.noise)


class FastDetector(Detector):
    """Same constructor and iteration protocol as `Detector`; `settings.carrier_thresh` and
    `settings.corr_thresh` are (const, snr, 0) in the power domain."""

    def __init__(self, settings, blocks=None, rxid=-1, yield_data=False, batch_size=None,
                 device_id=0):
        if yield_data:
            raise NotImplementedError("yield_data is not available in the fastdet variant")
        super(FastDetector, self).__init__(settings, blocks, rxid, False, batch_size=batch_size,
                                           device_id=device_id, _fastdet=True)


def _main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("input", type=argparse.FileType("rb"), default="-", nargs="?",
                        help="input data ('-' streams from stdin)")
    parser.add_argument("-o", "--output", type=argparse.FileType("w"), default=sys.stdout,
                        help="output file (.toad) [default: stdout]")
    parser.add_argument("--raw", action="store_true", help="input is raw u8 I/Q instead of .card")
    parser.add_argument("--tpl", type=str, default=None,
                        help="template in fastdet's .tpl format (overrides the .npy `template` setting)")
    keys = ["sample_rate", "block_size", "block_history", "carrier_window", "carrier_threshold",
            "corr_threshold", "template", "rxid"]
    config, args = load_args(parser, keys, argv=argv)
    template = load_tpl(args["tpl"]) if args["tpl"] else np.load(config.template)
    window = normalize_freq_range(config.carrier_window, config.sample_rate / config.block_size)
    settings = DetectorSettings(config.block_size, config.block_history, len(template),
                                config.carrier_threshold, window, template, config.corr_threshold)
    blocks = (RawStream(args["input"], config.block_size, config.block_history) if args["raw"]
              else CardStream(args["input"], config.block_size))
    det = FastDetector(settings, blocks, rxid=config.rxid)
    det.only_detections = True
    for detected, result in det:
        if detected:
            print(fastdet_line(result), file=args["output"])
    args["output"].flush()


if __name__ == "__main__":
    _main()
