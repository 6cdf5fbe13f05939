#!/usr/bin/env python3
"""Golden fixtures for the PreshiftDetector variant, made by
RUNNING THE REFERENCE's ``thrifty.experimental.detect_preshift.PreshiftDetector`` over the
input blocks of existing fixtures.  Build container only (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_preshift.py

Stores inputs + the reference's numeric outputs only.
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

from thrifty import block_data  # noqa: E402
from thrifty.detect import DetectorSettings  # noqa: E402
from thrifty.experimental import carrier_interpolators  # noqa: E402
from thrifty.experimental.detect_preshift import PreshiftDetector  # noqa: E402
from thrifty.signal_utils import Signal  # noqa: E402


# what run() writes (tests/test_oracle_golden.py checks the committed files against these lists):
# every fixture KEYS; those of the default interpolator carry their input blocks, the others name
# the fixture whose blocks they share
KEYS = ["block_len", "history_len", "carrier_thresh", "carrier_window", "corr_thresh", "template", "rxid", "num",
        "interpolator", "block_idx", "toad", "versions",
        "carrier_det", "det", "cbin", "coff", "cenergy", "cnoise", "sample", "soff", "energy", "noise", "soa",
        "frac_shift", "index_error"]
KEYS_OWN_BLOCKS = ["blocks"]
KEYS_SHARED_BLOCKS = ["src"]


def run(src_name, out_name, num, take=None, interpolator="parabolic"):
    g = np.load(os.path.join(HERE, src_name + ".npz"))
    st = DetectorSettings(int(g["block_len"]), int(g["history_len"]), len(g["template"]),
                          tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                          g["template"], tuple(g["corr_thresh"]))
    blocks, idx = g["blocks"], g["block_idx"]
    if take is not None:
        blocks, idx = blocks[take], idx[take]
    nb = len(blocks)
    rxid = int(g["rxid"])
    det = PreshiftDetector(st, None, rxid=rxid, num=num,
                           interpolator=carrier_interpolators.INTERPOLATORS[interpolator])
    out = {
        "carrier_det": np.zeros(nb, bool), "det": np.zeros(nb, bool),
        "cbin": np.zeros(nb, np.int64), "coff": np.zeros(nb, np.float32),   # (`none` returns the int 0)
        "cenergy": np.zeros(nb, np.float32), "cnoise": np.zeros(nb, np.float32),
        "sample": np.full(nb, -1, np.int64), "soff": np.zeros(nb),
        "energy": np.zeros(nb), "noise": np.zeros(nb), "soa": np.full(nb, np.nan),
        "frac_shift": np.zeros(nb), "index_error": np.zeros(nb, bool),
    }
    lines = []
    for i in range(nb):
        sig = Signal(block_data.raw_to_complex(blocks[i]))
        try:
            detected, res = det.detect(1000.0 + i, int(idx[i]), sig)
        except IndexError:
            out["index_error"][i] = True
            _, pk, _, _ = det.sync.detector(sig.fft.mag)
            out["cbin"][i] = pk
            continue
        ci = res.carrier_info
        out["carrier_det"][i] = res.corr_info is not None
        out["det"][i] = detected
        out["cbin"][i], out["coff"][i] = ci.bin, ci.offset
        out["cenergy"][i], out["cnoise"][i] = ci.energy, ci.noise
        if res.corr_info is not None:
            co = res.corr_info
            out["sample"][i], out["soff"][i] = co.sample, co.offset
            out["energy"][i], out["noise"][i], out["soa"][i] = co.energy, co.noise, res.soa
            out["frac_shift"][i] = det.frac_shift
        if detected:
            lines.append(res.serialize())
    meta = dict(block_len=st.block_len, history_len=st.history_len,
                carrier_thresh=np.array(st.carrier_thresh, float),
                carrier_window=np.array(st.carrier_window, np.int64),
                corr_thresh=np.array(st.corr_thresh, float), template=np.asarray(st.template),
                rxid=rxid, num=num, interpolator=interpolator, blocks=blocks, block_idx=np.asarray(idx, np.int64),
                toad="\n".join(lines),
                versions="numpy %s scipy %s python %s" % (np.__version__, scipy.__version__,
                                                           sys.version.split()[0]))
    meta.update(out)
    if interpolator != "parabolic":
        # the input blocks are those of fixture `src_name` (whole, in order): not stored again
        assert take is None
        del meta["blocks"]
        meta["src"] = src_name
    assert sorted(meta) == sorted(KEYS + (KEYS_OWN_BLOCKS if interpolator == "parabolic" else KEYS_SHARED_BLOCKS))
    path = os.path.join(HERE, out_name + ".npz")
    np.savez_compressed(path, **meta)
    print("%-22s blocks=%d carrier=%d det=%d index_error=%d  %.0f KiB" % (
        out_name, nb, out["carrier_det"].sum(), out["det"].sum(), out["index_error"].sum(),
        os.path.getsize(path) / 1024))


if __name__ == "__main__":
    run("c2", "preshift_c2", 21)
    run("c2_straddle", "preshift_c2_straddle", 21)     # carrier bins around 0, negative bins
    run("c2_stddev", "preshift_c2_stddev", 11)         # stddev threshold terms, other bank size
    run("c1", "preshift_c1", 101, take=slice(0, 6))    # float64 template, CLI's --num 101
    run("small", "preshift_small", 21)                 # N = 4096
    # the reference's other three-point carrier interpolators (experimental/carrier_interpolators.py)
    for name in ("none", "gaussian", "cosine"):
        run("c2", "preshift_c2_" + name, 21, interpolator=name)
    run("c2_straddle", "preshift_c2_straddle_gaussian", 21, interpolator="gaussian")
