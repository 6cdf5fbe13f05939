#!/usr/bin/env python3
"""Golden fixtures for a REPLACED carrier interpolator (SURVEY.md 2 #10's users of
`Detector.sync.interpolator`), made by RUNNING THE REFERENCE's
``thrifty.experimental.detect_carrier_interpol.InterpolationDetector`` -- which assigns
``self.sync.interpolator`` (detect_carrier_interpol.py:17-40) -- over the input blocks of the `c2`
fixture.  Build container only (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_interpol.py

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
builtins.basestring = str          # (detect_carrier_interpol.py:23 is Python 2)

from thrifty import block_data  # noqa: E402
from thrifty.detect import DetectorSettings  # noqa: E402
from thrifty.experimental import carrier_interpolators  # noqa: E402
from thrifty.experimental.detect_carrier_interpol import InterpolationDetector  # noqa: E402
from thrifty.signal_utils import Signal  # noqa: E402

# what run() writes (tests/test_oracle_golden.py checks the committed files against this list)
KEYS = ["src", "method", "block_idx", "toad", "versions", "carrier_det", "det", "cbin", "coff", "coff_is_int",
        "cenergy", "cnoise", "sample", "soff", "energy", "noise", "soa"]

# name -> what is passed as `method` (a name of the reference's table, or a callable)
METHODS = {
    "none": "none", "parabolic": "parabolic", "gaussian": "gaussian", "cosine": "cosine",
    "parabole_fit6": lambda: carrier_interpolators.make_parabole_fit(6),
    "corr_parabolic4": lambda n, w: carrier_interpolators.make_corr_parabolic(4, n, w),
}


def run(src_name, name):
    g = np.load(os.path.join(HERE, src_name + ".npz"))
    st = DetectorSettings(int(g["block_len"]), int(g["history_len"]), len(g["template"]),
                          tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                          g["template"], tuple(g["corr_thresh"]))
    method = METHODS[name]
    if callable(method):
        method = method() if name == "parabole_fit6" else method(st.block_len, st.carrier_len)
    blocks, idx = g["blocks"], g["block_idx"]
    nb = len(blocks)
    det = InterpolationDetector(st, None, rxid=int(g["rxid"]), method=method)
    out = {
        "carrier_det": np.zeros(nb, bool), "det": np.zeros(nb, bool), "cbin": np.zeros(nb, np.int64),
        "coff": np.zeros(nb), "coff_is_int": np.zeros(nb, bool),
        "cenergy": np.zeros(nb, np.float32), "cnoise": np.zeros(nb, np.float32),
        "sample": np.full(nb, -1, np.int64), "soff": np.zeros(nb), "energy": np.zeros(nb), "noise": np.zeros(nb),
        "soa": np.full(nb, np.nan),
    }
    lines = []
    for i in range(nb):
        detected, res = det.detect(1000.0 + i, int(idx[i]), Signal(block_data.raw_to_complex(blocks[i])))
        ci = res.carrier_info
        out["carrier_det"][i] = res.corr_info is not None
        out["det"][i] = detected
        out["cbin"][i], out["coff"][i], out["coff_is_int"][i] = ci.bin, ci.offset, isinstance(ci.offset, int)
        out["cenergy"][i], out["cnoise"][i] = ci.energy, ci.noise
        if res.corr_info is not None:
            co = res.corr_info
            out["sample"][i], out["soff"][i] = co.sample, co.offset
            out["energy"][i], out["noise"][i], out["soa"][i] = co.energy, co.noise, res.soa
        if detected:
            lines.append(res.serialize())
    meta = dict(src=src_name, method=name, block_idx=np.asarray(idx, np.int64), toad="\n".join(lines),
                versions="numpy %s scipy %s python %s" % (np.__version__, scipy.__version__, sys.version.split()[0]))
    meta.update(out)
    assert sorted(meta) == sorted(KEYS)
    path = os.path.join(HERE, "interpol_%s_%s.npz" % (src_name, name))
    np.savez_compressed(path, **meta)
    print("%-32s blocks=%d carrier=%d det=%d int offsets=%d  %.0f KiB" % (
        os.path.basename(path), nb, out["carrier_det"].sum(), out["det"].sum(), out["coff_is_int"].sum(),
        os.path.getsize(path) / 1024))


if __name__ == "__main__":
    for name in METHODS:
        run("c2", name)
