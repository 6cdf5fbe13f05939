#!/usr/bin/env python3
"""Golden fixtures for the `identify` step, made by RUNNING THE
REFERENCE's thrifty/identify.py functions on synthetic detection sets.  Build container
only (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_identify.py

The reference module is Python-2 era (`dict.iteritems`): the dictionaries handed to it here
are subclasses that provide `iteritems`; nothing else is patched.  Stores the detections'
columns (inputs) and the reference's txids / window edges / duplicate masks / output order.
"""
import builtins
import collections
import os
import sys

import numpy as np

REF = os.environ.get("THRIFTY_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
builtins.xrange = range


class Dict2(dict):
    iteritems = dict.items


class DefaultDict2(collections.defaultdict):
    iteritems = collections.defaultdict.items


from thrifty import identify, toads_data  # noqa: E402

identify.defaultdict = DefaultDict2


def make_detections(rng, n_rx, tx_bins, n_events, dup_prob, stray_prob, jitter):
    """Per RX: n_events transmissions cycling over the TXs; a transmission may also trigger the
    block before/after with lower energy (the duplicates `identify` removes); some strays."""
    dets = []
    for rx in range(n_rx):
        rx_off = int(rng.integers(-6, 7))
        t0 = 1.7e9 + rx * 0.013
        for ev in range(n_events):
            tx = ev % len(tx_bins)
            block = 10 + 3 * ev + int(rng.integers(0, 2))
            cbin = tx_bins[tx] + rx_off + int(np.round(rng.normal(0, jitter)))
            coff = float(rng.uniform(-0.5, 0.5))
            energy = float(rng.uniform(80, 200))
            blocks = [(block, energy)]
            u = rng.random()
            if u < dup_prob:
                blocks.append((block + 1, energy * float(rng.uniform(0.2, 0.9))))
            elif u < 2 * dup_prob:
                blocks.append((block - 1, energy * float(rng.uniform(0.2, 0.9))))
            for blk, en in blocks:
                dets.append((rx, t0 + blk * 0.00512 + float(rng.uniform(0, 1e-4)), blk, cbin, coff, en))
        for _ in range(int(stray_prob * n_events)):
            blk = int(rng.integers(0, 3 * n_events))
            dets.append((rx, t0 + blk * 0.00512, blk, int(rng.integers(min(tx_bins) - 30, max(tx_bins) + 30)),
                         float(rng.uniform(-0.5, 0.5)), float(rng.uniform(20, 60))))
    order = rng.permutation(len(dets))
    return [dets[i] for i in order]


def as_objects(rows):
    out = []
    for rx, ts, blk, cbin, coff, en in rows:
        car = toads_data.CarrierSyncInfo(cbin, coff, 150.0, 7.5)
        cor = toads_data.CorrDetectionInfo(4000 + blk % 97, 0.01 * (blk % 7), en, 1.5)
        out.append(toads_data.DetectionResult(ts, blk, 12288.0 * blk + cor.sample + cor.offset, car, cor, rx))
    return out


# what save() writes (tests/test_oracle_golden.py): the columns, plus the automatic windows or the map
KEYS = ["rxid", "timestamp", "block", "carrier_bin", "carrier_offset", "energy", "txid", "dup_mask", "kept_order"]
KEYS_AUTO = ["edge_rx", "edge_ptr", "edges"]
KEYS_MAP = ["map_rx", "map_tx", "map_lo", "map_hi"]


def save(name, rows, freqmap):
    dets = as_objects(rows)
    out = {}
    if freqmap is None:
        per_rx = {}
        for rx in sorted({r[0] for r in rows}):
            freqs = np.array([d.carrier_info.bin for d in dets if d.rxid == rx])
            per_rx[rx] = identify.detect_transmitter_windows(freqs)
        out["edge_rx"] = np.array(sorted(per_rx), np.int64)
        out["edge_ptr"] = np.cumsum([0] + [len(per_rx[rx]) for rx in sorted(per_rx)])
        out["edges"] = np.concatenate([per_rx[rx] for rx in sorted(per_rx)]).astype(np.int64)
        identify.identify_transmitters(dets, None)
    else:
        fm = Dict2({rx: Dict2(m) for rx, m in freqmap.items()})
        identify.identify_transmitters(dets, fm)
        out["map_rx"] = np.array([rx for rx in freqmap for _ in freqmap[rx]], np.int64)
        out["map_tx"] = np.array([tx for rx in freqmap for tx in freqmap[rx]], np.int64)
        out["map_lo"] = np.array([freqmap[rx][tx][0] for rx in freqmap for tx in freqmap[rx]], float)
        out["map_hi"] = np.array([freqmap[rx][tx][1] for rx in freqmap for tx in freqmap[rx]], float)
    txid = np.array([d.txid for d in dets], np.int64)
    mask = identify.identify_duplicates(dets)
    kept = identify.filter_duplicates(dets)
    index_of = {id(d): i for i, d in enumerate(dets)}
    cols = np.array(rows, dtype=float)
    assert sorted(out) == sorted(KEYS_AUTO if freqmap is None else KEYS_MAP)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        rxid=cols[:, 0].astype(np.int64), timestamp=cols[:, 1],
                        block=cols[:, 2].astype(np.int64), carrier_bin=cols[:, 3].astype(np.int64),
                        carrier_offset=cols[:, 4], energy=cols[:, 5], txid=txid,
                        dup_mask=np.asarray(mask, bool),
                        kept_order=np.array([index_of[id(d)] for d in kept], np.int64), **out)
    print("%-20s n=%d kept=%d txids=%s" % (name, len(rows), len(kept), sorted(set(txid.tolist()))))


def main():
    rng = np.random.default_rng(20260928 + 30)
    save("identify_auto3", make_detections(rng, 2, [25, 52, 88], 400, 0.25, 0.0, 0.6), None)
    save("identify_auto1", make_detections(rng, 1, [40], 120, 0.3, 0.0, 0.5), None)
    rows = make_detections(rng, 3, [20, 45, 70, 95], 300, 0.2, 0.08, 0.8)
    freqmap = {rx: {tx: (b - 4.0 + rx, b + 4.0 + rx) for tx, b in enumerate([20, 45, 70, 95])}
               for rx in range(3)}
    save("identify_map", rows, freqmap)


if __name__ == "__main__":
    main()
