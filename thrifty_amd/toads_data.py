"""Detection records and the .toad text format.

Mirrors the reference's data model (thrifty/toads_data.py:8-90): field names and
order of `CarrierSyncInfo` / `CorrDetectionInfo` / `DetectionResult` are API, and
`serialize()` produces the same whitespace-separated line
(`[rxid] [txid] t block soa sample offset energy noise cbin coffset cenergy cnoise`,
floats printed with Python's shortest repr except t (%.6f) and soa (%.8f)).
"""
from __future__ import annotations

from collections import namedtuple

import numpy as np

CarrierSyncInfo = namedtuple("CarrierSyncInfo", ["bin", "offset", "energy", "noise"])
CorrDetectionInfo = namedtuple("CorrDetectionInfo", ["sample", "offset", "energy", "noise"])



try:
    # the C base of DetectionResult (csrc/fastresults.c): the seven attributes, and a constructor
    # Detector uses to make a whole batch of results from the engine's records in one call
    from thrifty_amd._fastresults import ResultBase as _ResultBase
except ImportError:        # library not built (python -m thrifty_amd.build): the same object in Python
    class _ResultBase(object):
        __slots__ = ("timestamp", "block", "soa", "carrier_info", "corr_info", "rxid", "txid")

        def __init__(self, timestamp, block, soa, carrier_info, corr_info, rxid=None, txid=None):
            for name, value in zip(self.__slots__, (timestamp, block, soa, carrier_info, corr_info, rxid, txid)):
                setattr(self, name, value)


class DetectionResult(_ResultBase):
    """One block's verdict: timestamp, block index, SoA and both info tuples
    (`DetectionResult(timestamp, block, soa, carrier_info, corr_info, rxid=None, txid=None)`,
    reference toads_data.py:22-45)."""

    __slots__ = ()

    def serialize(self):
        # (a result the detector built from an engine record, untouched since: the engine library's
        # line for that record -- the same text, formatted in C)
        fast = getattr(self, "_serialize_fast", None)
        if fast is not None:
            text = fast()
            if text is not None:
                return text
        columns = ["%d" % v for v in (self.rxid, self.txid) if v is not None]
        columns += ["%.6f" % self.timestamp, "%d" % self.block, "%.8f" % self.soa]
        columns += map("{}".format, tuple(self.corr_info) + tuple(self.carrier_info))
        return " ".join(columns)

    @classmethod
    def deserialize(cls, string, with_rxid=False, with_txid=False):
        n_ids = bool(with_rxid) + bool(with_txid)
        parts = string.split()
        if len(parts) < n_ids + 11:
            return None
        ids = iter([int(p) for p in parts[:n_ids]])
        num = [float(p) for p in parts[n_ids:n_ids + 11]]
        return cls(num[0], int(num[1]), num[2], CarrierSyncInfo(int(num[7]), *num[8:11]),
                   CorrDetectionInfo(int(num[3]), *num[4:7]),
                   rxid=next(ids) if with_rxid else None, txid=next(ids) if with_txid else None)

    def __repr__(self):
        return "DetectionResult(block=%r, soa=%r, carrier=%r, corr=%r)" % (
            self.block, self.soa, self.carrier_info, self.corr_info)


def toad_lines(recs, timestamps, new_len, rxid=None, txid=None, carrier_offset_type=float):
    """`DetectionResult.serialize()` for a whole batch of engine records at once.

    recs: structured records (thr_record layout) of DETECTED blocks; timestamps: one float each.
    Whole columns are converted together -- `repr` over `.tolist()` for the float fields (the
    carrier energy / noise are np.float32 in the reference, and `'{}'.format(np.float32)` prints
    the value widened to a double, so they too are the repr of the widened value), `%.6f` /
    `%.8f` for timestamp / soa -- and joined, with no per-record result objects.  Text is identical to building each `DetectionResult` and
    serialising it (tests/test_host_logic.py)."""
    n = len(recs)
    if n == 0:
        return []
    block = recs["block_idx"].astype(np.int64)
    sample = recs["corr_sample"].astype(np.int64)
    off = recs["corr_offset"].astype(np.float64)
    # soa = new_len * block_idx + sample + offset: exact integer part, then one float64 add
    soa = (np.int64(new_len) * block + sample).astype(np.float64) + off
    cols = [
        ["%.6f" % v for v in np.asarray(timestamps, dtype=np.float64).tolist()],
        list(map(str, block.tolist())),
        ["%.8f" % v for v in soa.tolist()],
        list(map(str, sample.tolist())),
        list(map(repr, off.tolist())),
        list(map(repr, recs["corr_energy"].astype(np.float64).tolist())),
        list(map(repr, recs["corr_noise"].astype(np.float64).tolist())),
        list(map(str, recs["carrier_bin"].tolist())),
        # (PreshiftDetector: np.float32 in the reference -- rounded to it, printed widened)
        list(map(repr, recs["carrier_offset"].astype(carrier_offset_type).astype(np.float64).tolist())),
        list(map(repr, recs["carrier_energy"].astype(np.float64).tolist())),
        list(map(repr, recs["carrier_noise"].astype(np.float64).tolist())),
    ]
    # THR_FLAG_INT_OFFSET (8): the reference's interpolator returned the Python int 0 for this block
    # (cosine, cos(omega) > 1, carrier_interpolators.py:87-88) -- the column reads "0"
    for i in np.flatnonzero(recs["flags"] & 8).tolist():
        cols[8][i] = "0"
    ids = [str(v) for v in (rxid, txid) if v is not None]
    head = " ".join(ids) + " " if ids else ""
    return [head + " ".join(row) for row in zip(*cols)]


def _parsed_lines(lines, with_rxid, with_txid):
    """(line number, DetectionResult or None) of every line that is not a comment."""
    for lineno, line in enumerate(lines, 1):
        text = line.decode() if isinstance(line, bytes) else line
        if text[:1] not in ("", "#"):
            yield lineno, DetectionResult.deserialize(text, with_rxid=with_rxid, with_txid=with_txid)


def _read(stream, with_rxid, with_txid):
    if isinstance(stream, str):
        with open(stream, "r") as handle:
            return _read(handle, with_rxid, with_txid)
    parsed = list(_parsed_lines(stream, with_rxid, with_txid))
    for lineno in (n for n, rec in parsed if rec is None):
        print("WARNING: skipped line #{}: line's formatting is invalid".format(lineno))
    return [rec for _, rec in parsed if rec is not None]


def load_toad(stream):
    """Single receiver's detections (.toad: rxid, no txid)."""
    return _read(stream, True, False)


def load_toads(stream):
    """Merged detections (.toads: rxid and txid)."""
    return _read(stream, True, True)


TOADS_DTYPE = [("idx", "i4"), ("rxid", "i4"), ("txid", "i4"), ("timestamp", "f8"),
               ("block", "i4"), ("soa", "f8"), ("sample", "i4"), ("offset", "f8"),
               ("energy", "f8"), ("noise", "f8"), ("carrier_bin", "i4"),
               ("carrier_offset", "f8"), ("carrier_energy", "f8"), ("carrier_noise", "f8")]


def toads_array(detections, with_ids=True):
    def _id(v):
        return -1 if (v is None or not with_ids) else v

    rows = [(i, _id(d.rxid), _id(d.txid), d.timestamp, d.block,
             d.soa, d.corr_info.sample, d.corr_info.offset, d.corr_info.energy,
             d.corr_info.noise, d.carrier_info.bin, d.carrier_info.offset,
             d.carrier_info.energy, d.carrier_info.noise) for i, d in enumerate(detections)]
    return np.array(rows, dtype=TOADS_DTYPE)
