"""Dev/aux: list the blocks of one tests/test_gpu_soak.py case with the largest deviations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import soak_util
import test_gpu_soak as T
from oracle import thrifty_np as onp
from thrifty_amd import _native as F, synth

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "full_spectrum_window"
    case = [c for c in T.CASES if c[0] == name][0]
    _, nb, cthr, cwin, xthr, what = case
    rng = np.random.default_rng(4242 + [c[0] for c in T.CASES].index(name))
    tpl = synth.gold_template(10, 3).astype(np.float64)
    win = onp.unique_window(T.N, T.H, len(tpl))
    lo_bin, hi_bin = (10.0, 100.0) if cwin[0] >= 0 else (-35.0, 55.0)
    blocks, truth = synth.synth_blocks(rng, nb, T.N, tpl, win, signal_frac=0.85, carrier_bins=(lo_bin, hi_bin))
    eng = F.Engine(T.N, T.H, tpl, cthr, cwin, xthr, max_batch=1024)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    rows = soak_util.run_oracle(blocks, T.N, T.H, tpl, cthr, cwin, xthr)
    dev = []
    for i, row in enumerate(rows):
        if row is None or not row[1]:
            continue
        r = rec[i]
        dev.append((abs(r["carrier_offset"] - row[2]), i))
    dev.sort(reverse=True)
    print("truth keys", list(truth.keys()))
    for d, i in dev[:12]:
        row, r = rows[i], rec[i]
        has = truth["has_signal"][i] if "has_signal" in truth else None
        print("blk %5d has=%s bin gpu %5d orc %5d coff gpu %+.6f orc %+.6f (d %.2e) cen %.3f | sample %d/%d energy %.5f/%.5f det %s/%s off %+.6f/%+.6f" % (
            i, has, r["carrier_bin"], row[0], r["carrier_offset"], row[2], d, row[3], r["corr_sample"], row[4],
            r["corr_energy"], row[6], bool(r["flags"] & 2), row[5], r["corr_offset"], row[7]))
