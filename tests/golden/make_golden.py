#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, read-only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

It imports the reference's own ``thrifty.detect.Detector`` (np.fft branch; no
pyFFTW here), feeds it seeded synthetic u8 IQ blocks and
stores inputs + the reference's outputs/intermediates as .npz data, plus one
small .card stream with the .toad text the reference prints for it.  Nothing
of the reference's source is stored -- only inputs and numeric outputs.
"""
import base64
import builtins
import io
import os
import sys

import numpy as np
import scipy

REF = os.environ.get("THRIFTY_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
builtins.xrange = range  # gold.py:77 is Python-2 era

from thrifty import block_data, gold, template_generate  # noqa: E402
from thrifty.detect import Detector, DetectorSettings  # noqa: E402
from thrifty.signal_utils import Signal  # noqa: E402

from thrifty_amd import synth  # noqa: E402  (input generator only)

SEED0 = 20260928

# what save() writes into every fixture (tests/test_oracle_golden.py checks the committed files
# against this list; `small` adds the .card round trip)
KEYS = ["block_len", "history_len", "carrier_thresh", "carrier_window", "corr_thresh", "template", "rxid",
        "blocks", "block_idx", "toad", "versions",
        "carrier_det", "det", "cbin", "coff", "cenergy", "cnoise", "sample", "soff", "energy", "noise", "soa",
        "sum_mag2", "nbhd", "xhat_energy", "corr3", "index_error"]
EXTRA_KEYS = {"small": ["card_text", "card_toad"]}


def run_reference(settings, blocks_u8, block_idx, rxid=0):
    det = Detector(settings, None, rxid=rxid, yield_data=True)
    nb = len(blocks_u8)
    out = {
        "carrier_det": np.zeros(nb, bool), "det": np.zeros(nb, bool),
        "cbin": np.zeros(nb, np.int64), "coff": np.zeros(nb),
        "cenergy": np.zeros(nb, np.float32), "cnoise": np.zeros(nb, np.float32),
        "sample": np.full(nb, -1, np.int64), "soff": np.zeros(nb),
        "energy": np.zeros(nb), "noise": np.zeros(nb), "soa": np.full(nb, np.nan),
        "sum_mag2": np.zeros(nb, np.float32), "nbhd": np.zeros((nb, 7), np.float32),
        "xhat_energy": np.zeros(nb), "corr3": np.zeros((nb, 3)),
        "index_error": np.zeros(nb, bool),
    }
    lines = []
    for i in range(nb):
        sig = Signal(block_data.raw_to_complex(blocks_u8[i]))
        try:
            detected, res, xhat, corr = det.detect(1000.0 + i, int(block_idx[i]), sig)
        except IndexError:
            # carrier_sync.py:187: peak_idx + 3 >= N is not wrapped by the reference
            out["index_error"][i] = True
            _, pk, _, _ = det.sync.detector(sig.fft.mag)
            out["cbin"][i] = pk
            continue
        mag = sig.fft.mag
        ci = res.carrier_info
        out["carrier_det"][i] = res.corr_info is not None
        out["det"][i] = detected
        out["cbin"][i] = ci.bin
        out["coff"][i] = ci.offset
        out["cenergy"][i] = ci.energy
        out["cnoise"][i] = ci.noise
        out["sum_mag2"][i] = np.sum(mag ** 2)
        out["nbhd"][i] = mag[(ci.bin + np.arange(-3, 4)) % len(mag)]
        if res.corr_info is not None:
            co = res.corr_info
            out["sample"][i] = co.sample
            out["soff"][i] = co.offset
            out["energy"][i] = co.energy
            out["noise"][i] = co.noise
            out["soa"][i] = res.soa
            out["xhat_energy"][i] = np.mean(np.abs(xhat) ** 2)
            cm = np.abs(corr)
            out["corr3"][i] = cm[co.sample - 1:co.sample + 2] if 0 < co.sample < len(cm) - 1 else 0
        if detected:
            lines.append(res.serialize())
    return out, lines


def save(name, settings, blocks_u8, block_idx, extra=None, rxid=0):
    out, lines = run_reference(settings, blocks_u8, block_idx, rxid)
    meta = dict(
        block_len=settings.block_len, history_len=settings.history_len,
        carrier_thresh=np.array(settings.carrier_thresh, float),
        carrier_window=np.array(settings.carrier_window, np.int64),
        corr_thresh=np.array(settings.corr_thresh, float),
        template=np.asarray(settings.template), rxid=rxid,
        blocks=blocks_u8, block_idx=np.asarray(block_idx, np.int64),
        toad="\n".join(lines),
        versions="numpy %s scipy %s python %s" % (np.__version__, scipy.__version__,
                                                   sys.version.split()[0]),
    )
    meta.update(out)
    if extra:
        meta.update(extra)
    assert sorted(meta) == sorted(KEYS + EXTRA_KEYS.get(name, [])), sorted(meta)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **meta)
    print("%-18s blocks=%d carrier=%d det=%d  %.0f KiB" % (
        name, len(blocks_u8), out["carrier_det"].sum(), out["det"].sum(),
        os.path.getsize(path) / 1024))
    return out


def window_of(n, h, w):
    pad = h - w + 1
    left = pad // 2
    return left, (n - w + 1) - (pad - left)


def mixed_blocks(rng, n, template, win, count):
    """Signal / noise-only / carrier-without-code / window-edge / overlap mix."""
    w = len(template)
    lo, hi = win
    blocks, truth = synth.synth_blocks(rng, count, n, template, win)
    k = count - 8
    # noise only
    blocks[k:k + 2], _ = synth.synth_blocks(rng, 2, n, template, win, signal_frac=0.0)
    # unmodulated carrier burst (carrier detects, code does not)
    ones = np.ones(w)
    blocks[k + 2:k + 4], _ = synth.synth_blocks(rng, 2, n, ones, win, amp=0.15)
    # peaks on both edges of the unique window, and one lag outside each edge
    blocks[k + 4:k + 8], _ = synth.synth_blocks(
        rng, 4, n, template, win, positions=[lo, hi - 1, lo - 1, hi])
    return blocks


def main():
    # ---- C2: N=16384 H=4096, 10-bit Gold @ 1 sps ------------------------
    rng = np.random.default_rng(SEED0 + 2)
    n, h = 16384, 4096
    tpl = template_generate.resample(gold.gold(10, 2), 1.0)
    win = window_of(n, h, len(tpl))
    st = DetectorSettings(n, h, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
    blocks = mixed_blocks(rng, n, tpl, win, 24)
    save("c2", st, blocks, np.arange(24) * 3 + 5)

    # ---- C2 variants: negative-bin window, straddling window, stddev term -
    rng = np.random.default_rng(SEED0 + 20)
    blocks, _ = synth.synth_blocks(rng, 6, n, tpl, win, carrier_bins=(-90.0, -20.0))
    st_neg = st._replace(carrier_window=(-110, -7))
    save("c2_negwin", st_neg, blocks, np.arange(6))
    blocks, _ = synth.synth_blocks(rng, 6, n, tpl, win, carriers=[-8.3, -0.4, 0.0, 0.45, 3.2, 9.1])
    st_str = st._replace(carrier_window=(-12, 12))
    save("c2_straddle", st_str, blocks, np.arange(6))
    blocks, _ = synth.synth_blocks(rng, 6, n, tpl, win)
    st_std = st._replace(carrier_thresh=(100.0, 5.0, 2.0), corr_thresh=(50.0, 8.0, 3.0))
    save("c2_stddev", st_std, blocks, np.arange(6))
    blocks, _ = synth.synth_blocks(rng, 6, n, tpl, win, carrier_bins=(-4000.0, 4000.0))
    st_full = st._replace(carrier_window=(0, -1))
    save("c2_fullwin", st_full, blocks, np.arange(6))

    # ---- multi-template: 4 TX codes, each block carries one of them ------
    rng = np.random.default_rng(SEED0 + 5)
    tpls = [template_generate.resample(gold.gold(10, i), 1.0) for i in (2, 3, 4, 5)]
    parts = []
    for t in tpls:
        b, _ = synth.synth_blocks(rng, 3, n, t, win)
        parts.append(b)
    blocks = np.concatenate(parts)
    for ti, t in enumerate(tpls):
        save("c5_tx%d" % ti, st._replace(template=t), blocks, np.arange(12))

    # ---- C1: example/detector.cfg + example/template.npy ----------------
    rng = np.random.default_rng(SEED0 + 1)
    tpl1 = np.load(os.path.join(REF, "example", "template.npy"))
    n, h = 16384, 4920
    win = window_of(n, h, len(tpl1))
    st1 = DetectorSettings(n, h, len(tpl1), (0.0, 15.0, 0.0), (7, 110), tpl1, (0.0, 15.0, 0.0))
    blocks = mixed_blocks(rng, n, tpl1 / np.max(np.abs(tpl1)), win, 12)
    save("c1", st1, blocks, np.arange(12) + 100)

    # ---- C3: N=65536 H=4096, 11-bit Gold @ 2 sps -------------------------
    rng = np.random.default_rng(SEED0 + 3)
    n, h = 65536, 4096
    tpl3 = template_generate.resample(gold.gold(11, 2), 2.0)
    win = window_of(n, h, len(tpl3))
    st3 = DetectorSettings(n, h, len(tpl3), (0, 15, 0), (7, 110), tpl3, (0, 15, 0))
    blocks, _ = synth.synth_blocks(rng, 4, n, tpl3, win)
    save("c3", st3, blocks, np.arange(4))

    # ---- small: N=4096 H=1024, 8-bit Gold @ 2 sps (+ .card/.toad text) ---
    rng = np.random.default_rng(SEED0 + 9)
    n, h = 4096, 1024
    tpls = template_generate.resample(gold.gold(8, 3), 2.0)
    win = window_of(n, h, len(tpls))
    sts = DetectorSettings(n, h, len(tpls), (0, 12, 0), (5, 60), tpls, (0, 12, 0))
    blocks = mixed_blocks(rng, n, tpls, win, 14)
    idx = np.arange(14) * 2 + 40
    card = io.StringIO()
    card.write("# arguments: { synthetic }\nUsing Volk machine: avx2_64_mmx\n\n")
    for i in range(14):
        card.write("%d.%06d %d %s\n" % (1475000000 + i, 123456 + 7 * i, idx[i],
                                        base64.b64encode(blocks[i].tobytes()).decode()))
    card_text = card.getvalue()
    # run the reference's own card_reader over the text, then its Detector
    det = Detector(sts, block_data.card_reader(io.StringIO(card_text)), rxid=3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # np.fromstring deprecation (block_data.py:129)
        toad = [res.serialize() for detected, res in det if detected]
    save("small", sts, blocks, idx, extra={"card_text": card_text,
                                            "card_toad": "\n".join(toad)}, rxid=3)


if __name__ == "__main__":
    main()
