#!/usr/bin/env python
"""
Merge RX detections, identify transmitter IDs, filter detections.

GPU counterpart of reference thrifty/identify.py: merges .toad
files, classifies every detection's transmitter from its carrier frequency (frequency map,
or automatic windows from the histogram of carrier bins), removes the duplicate detections
neighbouring blocks produce, and writes a .toads file.  Classification, the four-key sort,
the neighbour test and the output ordering run on the device (`thr_identify`,
csrc/identify.hip); this module keeps the reference's function names on top of it.
"""
from __future__ import print_function

import argparse
import glob

import numpy as np

from thrifty_amd import _native, toads_data
from thrifty_amd.settings import parse_kvconfig

UNIDENTIFIED = -1


def _columns(detections):
    n = len(detections)
    cols = {"rxid": np.empty(n, np.int32), "block": np.empty(n, np.int32),
            "timestamp": np.empty(n, np.float64), "carrier_bin": np.empty(n, np.int32),
            "carrier_offset": np.empty(n, np.float64), "energy": np.empty(n, np.float64)}
    for i, d in enumerate(detections):
        cols["rxid"][i] = -1 if d.rxid is None else d.rxid
        cols["block"][i] = d.block
        cols["timestamp"][i] = d.timestamp
        cols["carrier_bin"][i] = d.carrier_info.bin
        cols["carrier_offset"][i] = d.carrier_info.offset
        cols["energy"][i] = d.corr_info.energy
    return cols


def _flatten(freqmap):
    """{rxid: {txid: (lo, hi)}} -> rows (rxid, txid, lo, hi) in iteration order."""
    if freqmap is None:
        return None
    return [(rx, tx, lo, hi) for rx, ranges in freqmap.items() for tx, (lo, hi) in ranges.items()]


def integrate_columns(cols, freqmap=None, device_id=0):
    """-> (txid, keep mask, kept indices in output order) for detection columns."""
    return _native.identify(cols["rxid"], cols["block"], cols["timestamp"], cols["carrier_bin"],
                            cols["carrier_offset"], cols["energy"], _flatten(freqmap), device_id)


def identify_transmitters(detections, freqmap):
    """Set `.txid` of every detection in place (reference identify.py:124-137)."""
    txid, _, _ = integrate_columns(_columns(detections), freqmap)
    for det, tx in zip(detections, txid.tolist()):
        det.txid = tx


def identify_duplicates(detections):
    """Mask (True = keep) over detections that already carry a txid (identify.py:140-172)."""
    cols = _columns(detections)
    txid = np.array([UNIDENTIFIED if d.txid is None else d.txid for d in detections], np.int32)
    # classification is a pass-through here: one exact-match range per (rxid, txid) present
    return _mask_for(cols, txid)[0]


def _mask_for(cols, txid):
    # feed the known txids through the map path: freq := txid, ranges [tx, tx]
    pairs = sorted({(int(r), int(t)) for r, t in zip(cols["rxid"], txid) if t != UNIDENTIFIED})
    ranges = [(r, t, float(t), float(t)) for r, t in pairs] or [(0, 0, 0.5, 0.5)]
    _, keep, order = _native.identify(cols["rxid"], cols["block"], cols["timestamp"], txid,
                                      np.zeros(len(txid)), cols["energy"], ranges)
    return keep, order


def filter_duplicates(detections):
    """Detections without duplicates / unidentified ones, by timestamp (identify.py:175-181)."""
    cols = _columns(detections)
    txid = np.array([UNIDENTIFIED if d.txid is None else d.txid for d in detections], np.int32)
    _, order = _mask_for(cols, txid)
    return [detections[i] for i in order.tolist()]


def integrate(detections, freqmap=None):
    """Identify and filter in ONE device pass (identify.py:218-222)."""
    txid, _, order = integrate_columns(_columns(detections), freqmap)
    for det, tx in zip(detections, txid.tolist()):
        det.txid = tx
    return [detections[i] for i in order.tolist()]


def load_toad_files(toad_globs):
    filenames = [name for pattern in toad_globs for name in glob.glob(pattern)]
    return [det for name in filenames for det in toads_data.load_toad(name)], filenames


def load_freqmap(file_):
    """`<txid>: <lo> - <hi>` nominal ranges plus `@<rxid>: <offset>` per receiver
    (identify.py:184-215) -> {rxid: {txid: (lo + offset, hi + offset)}}."""
    if file_ is None:
        return None
    entries = parse_kvconfig(file_)
    offsets = {int(k[1:]): float(v) for k, v in entries.items() if k.startswith("@")}
    nominal = {int(k): tuple(float(x) for x in v.split("-")) for k, v in entries.items() if not k.startswith("@")}
    return {rx: {tx: (lo + off, hi + off) for tx, (lo, hi) in nominal.items()} for rx, off in offsets.items()}


_REMOVED = "Removed {} duplicates / unidentified transmisisons from {} detections."     # (the reference's sentence, sic)


def generate_toads(output, toad_globs, freqmap):
    detections, filenames = load_toad_files(toad_globs)
    kept = integrate(detections, freqmap)
    output.write("".join(["# source_files: [%s]\n" % " ".join(filenames)] + [d.serialize() + "\n" for d in kept]))
    print(_REMOVED.format(len(detections) - len(kept), len(detections)))


_CLI = (
    (("toad_file",), dict(type=str, nargs="*", default=["*.toad"], help="toad file(s) from receivers [default: *.toad]")),
    (("-o", "--output"), dict(type=argparse.FileType("w"), default="data.toads", help="output file [default: data.toads]")),
    (("-m", "--map"), dict(type=argparse.FileType("r"),
                           help="schema for mapping DFT index to transmitter ID [default: auto-detect]")),
)


def _main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for flags, options in _CLI:
        parser.add_argument(*flags, **options)
    args = parser.parse_args(argv)
    with args.output as out:
        generate_toads(out, args.toad_file, load_freqmap(args.map))


if __name__ == "__main__":
    _main()
