"""Dev/aux: randomised configuration fuzz -- random block length / history / template /
carrier window / thresholds / batch split, a few blocks each, GPU vs oracle.
Usage: fuzz_parity.py [n_configs] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import thrifty_np as onp
from thrifty_amd import _native as F, synth


def one(rng, k):
    n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 16384, 16384, 32768, 65536, 512]))
    kind = rng.integers(0, 3)
    if kind == 0:
        bits = int(rng.integers(6, 11))
        sps = float(rng.choice([1.0, 2.0]))
        tpl = synth.gold_template(bits, int(rng.integers(0, 5)), sps)
    elif kind == 1:
        tpl = rng.normal(0, 1, int(rng.integers(50, min(n // 4, 3000))))       # float template
    else:
        tpl = np.sign(rng.normal(0, 1, int(rng.integers(31, min(n // 4, 2000)))))
    w = len(tpl)
    if w > n // 2:
        tpl = tpl[:n // 2]; w = len(tpl)
    if os.environ.get("FUZZ_MIN_RATIO"):    # keep the Dirichlet fit well conditioned: W >= N / ratio
        need = n // int(os.environ["FUZZ_MIN_RATIO"])
        if w < need:
            tpl = np.resize(tpl, need) * np.sign(rng.normal(0, 1, need)); w = need
    h = int(rng.integers(w - 1, min(n - 2, w - 1 + n // 2)))
    lo = int(rng.integers(-n // 3, n // 3))
    width = int(rng.choice([5, 40, 100, 121, 122, 200, n // 8]))
    window = (lo, lo + width)
    if rng.random() < 0.15:
        window = (0, -1)
    cthr = (float(rng.choice([0, 50.0])), float(rng.choice([8, 15, 30])), float(rng.choice([0, 0, 1.5])))
    xthr = (float(rng.choice([0, 20.0])), float(rng.choice([8, 15])), float(rng.choice([0, 0, 2.0])))
    win = onp.unique_window(n, h, w)
    # carrier inside the window (signed bins)
    lo_b, hi_b = (min(window), max(window)) if window != (0, -1) else (-n // 4, n // 4)
    nb = 5
    blocks, truth = synth.synth_blocks(rng, nb, n, tpl / max(1e-9, np.max(np.abs(tpl))), win, signal_frac=0.8,
                                       carrier_bins=(lo_b + 0.3, hi_b - 0.3))
    # a window that contains bin 0 makes the quantiser's DC spike the "carrier" of a noise-only
    # block: a delta to which the Dirichlet-lobe fit is ill-conditioned (the reference's own
    # offset moves under a one-ulp change of its inputs, and the noise-only correlation peak that
    # follows it can flip): only bin and verdicts are compared on such blocks
    try:
        a, b = onp.window_to_indices(window[0], window[1], n)
        dc_in_window = a == 0 or b >= n       # starts at bin 0 or wraps past it
    except ValueError:
        dc_in_window = False                  # (both sides refuse the window below)
    desc = "n=%d h=%d w=%d kind=%d window=%s cthr=%s xthr=%s" % (n, h, w, kind, window, cthr, xthr)
    try:
        eng = F.Engine(n, h, tpl, cthr, window, xthr, max_batch=int(rng.integers(1, 7)))
    except F.NativeError as e:
        try:
            onp.window_to_indices(window[0], window[1], n)
        except Exception:
            return "refused-both", desc
        return "ENGINE-REFUSED " + str(e), desc
    fmt_c64 = rng.random() < 0.3
    inp = np.stack([onp.iq_u8_to_c64(b) for b in blocks]) if fmt_c64 else blocks
    rec = eng.detect(inp, np.arange(nb) + 3)[:, 0]
    orc = onp.OracleDetector(n, h, tpl, cthr, window, xthr)
    bad = []
    for i in range(nb):
        try:
            (res,) = orc.detect_u8(i + 3, blocks[i])
        except IndexError:
            if not rec[i]["flags"] & F.FLAG_INDEX_ERROR:
                bad.append("blk %d: oracle IndexError, gpu flags %x" % (i, rec[i]["flags"]))
            continue
        r = rec[i]
        if r["flags"] & F.FLAG_INDEX_ERROR:
            bad.append("blk %d: gpu INDEX_ERROR only" % i); continue
        if r["carrier_bin"] != res.carrier.bin:
            bad.append("blk %d: bin %d vs %d" % (i, r["carrier_bin"], res.carrier.bin)); continue
        if bool(r["flags"] & F.FLAG_CARRIER) != res.carrier.detected:
            bad.append("blk %d: carrier verdict (energy %g noise %g thr %g)" % (
                i, res.carrier.energy, res.carrier.noise, res.carrier.threshold)); continue
        if not res.carrier.detected:
            continue
        if dc_in_window and not truth["has_signal"][i]:
            continue
        if abs(r["carrier_offset"] - res.carrier.offset) > 1e-3:
            bad.append("blk %d: carrier offset %g vs %g" % (i, r["carrier_offset"], res.carrier.offset))
        if r["corr_sample"] != res.corr.sample:
            bad.append("blk %d: sample %d vs %d" % (i, r["corr_sample"], res.corr.sample)); continue
        if bool(r["flags"] & F.FLAG_CORR) != res.corr.detected:
            margin = abs(res.corr.energy - res.corr.threshold) / res.corr.threshold
            bad.append("blk %d: corr verdict (margin %.2g)" % (i, margin)); continue
        if abs(r["corr_energy"] - res.corr.energy) > 1e-4 * res.corr.energy:
            bad.append("blk %d: energy %g vs %g" % (i, r["corr_energy"], res.corr.energy))
        if res.corr.detected and abs(r["corr_offset"] - res.corr.offset) > 1e-4:
            bad.append("blk %d: offset %g vs %g" % (i, r["corr_offset"], res.corr.offset))
    return ("; ".join(bad) if bad else "ok"), desc + (" c64" if fmt_c64 else " u8")


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    tally = {}
    t0 = time.time()
    for k in range(count):
        try:
            status, desc = one(rng, k)
        except Exception as e:   # noqa
            status, desc = "EXCEPTION %r" % (e,), "config %d" % k
        key = status if status in ("ok", "refused-both") else "FAIL"
        tally[key] = tally.get(key, 0) + 1
        if key == "FAIL":
            print("[%d] %s\n      %s" % (k, desc, status))
    print("fuzz: %s in %.0f s" % (tally, time.time() - t0))


if __name__ == "__main__":
    main()
