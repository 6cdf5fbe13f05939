"""thrifty_amd -- MI355X (gfx950) matched-filter detection engine behind Thrifty's
`Detector` / `DetectorSettings` API and `.toad` format.

    from thrifty_amd.detect import Detector, DetectorSettings, detector_cli

The compute path is hand-written HIP in `libthriftyhip.so` (C ABI: include/thrifty_hip.h),
bound with ctypes in `thrifty_amd._native`.  Importing this package never needs a GPU;
constructing a `Detector`/`Engine` does, and fails loudly without one (no CPU fallback).
"""
__version__ = "0.1.0"
