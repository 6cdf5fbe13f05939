#!/usr/bin/env python3
"""Golden fixtures for a REPLACED correlation-peak interpolator (`Detector.soa_estimate.interpolate`),
made by RUNNING THE REFERENCE's ``thrifty.experimental.detect_xcorr_interpol.InterpolationDetector``
-- which assigns ``self.soa_estimate.interpolate`` or, for `maximise`, swaps in its
IterativeSoaEstimator (detect_xcorr_interpol.py:20-62) -- over the input blocks of the `c2` fixture.
Build container only (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_xcorr.py

Stores the reference's numeric outputs only; the blocks are those of fixture `src`.
"""
import builtins
import os
import sys

import numpy as np
import scipy

REF = os.environ.get("THRIFTY_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
builtins.xrange = range
builtins.basestring = str

from thrifty import block_data  # noqa: E402
from thrifty.detect import DetectorSettings  # noqa: E402
from thrifty.experimental.detect_xcorr_interpol import InterpolationDetector  # noqa: E402
from thrifty.signal_utils import Signal  # noqa: E402

# what run() writes (tests/test_oracle_golden.py checks the committed files against this list)
KEYS = ["src", "method", "block_idx", "toad", "versions", "carrier_det", "det", "cbin", "coff",
        "sample", "soff", "soff_is_int", "energy", "noise", "soa"]
# (fixture, method): `autocorr` scales a correlation of the template IN PLACE (xcorr_interpolators.py:68),
# which NumPy refuses for c2's integer template -- it runs on c1, whose template is float64
CASES = [("c2", "none"), ("c2", "parabolic"), ("c2", "cosine"), ("c2", "gaussian"), ("c2", "maximise"),
         ("c1", "autocorr"), ("c1", "maximise"), ("c1", "parabolic")]


def run(src_name, name):
    g = np.load(os.path.join(HERE, src_name + ".npz"))
    st = DetectorSettings(int(g["block_len"]), int(g["history_len"]), len(g["template"]),
                          tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                          g["template"], tuple(g["corr_thresh"]))
    blocks, idx = g["blocks"], g["block_idx"]
    nb = len(blocks)
    det = InterpolationDetector(st, None, rxid=int(g["rxid"]), method=name)
    out = {
        "carrier_det": np.zeros(nb, bool), "det": np.zeros(nb, bool), "cbin": np.zeros(nb, np.int64),
        "coff": np.zeros(nb), "sample": np.full(nb, -1, np.int64), "soff": np.zeros(nb),
        "soff_is_int": np.zeros(nb, bool), "energy": np.zeros(nb), "noise": np.zeros(nb), "soa": np.full(nb, np.nan),
    }
    lines = []
    for i in range(nb):
        detected, res = det.detect(1000.0 + i, int(idx[i]), Signal(block_data.raw_to_complex(blocks[i])))
        out["carrier_det"][i] = res.corr_info is not None
        out["det"][i] = detected
        out["cbin"][i], out["coff"][i] = res.carrier_info.bin, res.carrier_info.offset
        if res.corr_info is not None:
            co = res.corr_info
            out["sample"][i], out["soff"][i], out["soff_is_int"][i] = co.sample, co.offset, isinstance(co.offset, int)
            out["energy"][i], out["noise"][i], out["soa"][i] = co.energy, co.noise, res.soa
        if detected:
            lines.append(res.serialize())
    meta = dict(src=src_name, method=name, block_idx=np.asarray(idx, np.int64), toad="\n".join(lines),
                versions="numpy %s scipy %s python %s" % (np.__version__, scipy.__version__, sys.version.split()[0]))
    meta.update(out)
    assert sorted(meta) == sorted(KEYS)
    path = os.path.join(HERE, "xcorr_%s_%s.npz" % (src_name, name))
    np.savez_compressed(path, **meta)
    d = out["det"]
    print("%-28s blocks=%d det=%d int offsets=%d |offset| max %.3f clipped=%d  %.0f KiB" % (
        os.path.basename(path), nb, d.sum(), (out["soff_is_int"] & d).sum(), np.abs(out["soff"][d]).max(),
        (np.abs(out["soff"][d]) >= 0.6).sum(), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    for src, name in CASES:
        run(src, name)
