"""Dev/aux: THR_FLAG_FIT_UNCONVERGED against the reference's RuntimeError (carrier_sync.py:189: SciPy's
curve_fit gives up with lmdif exit code 5 .. 8) on a degenerate geometry -- a 64-sample template at
block_len 16384, where the seven fitted magnitudes sit on a flat main lobe.  Prints the blocks the
oracle raises on, the blocks the engine flags, and what they share.
Usage: fit_flag_probe.py [n_blocks] [seed] [history] [template_len]"""
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

N = 16384
THR = (0, 15, 0)


def make(seed, h, w, nb):
    from oracle import thrifty_np as onp
    from thrifty_amd import synth
    rng = np.random.default_rng(seed)
    tpl = np.sign(rng.normal(0, 1, w))
    blocks, _ = synth.synth_blocks(rng, nb, N, tpl, onp.unique_window(N, h, w), carrier_bins=(12.0, 100.0))
    return tpl, blocks


def _raises(job):
    from oracle import thrifty_np as onp
    h, tpl, blocks, first = job
    orc = onp.OracleDetector(N, h, tpl, THR, (7, 110), THR)
    out = []
    for i, raw in enumerate(blocks):
        try:
            orc.detect_u8(first + i, raw)
        except RuntimeError as exc:
            if "Optimal parameters not found" not in str(exc):
                raise
            out.append(first + i)
    return out


def reference_raises(h, tpl, blocks, procs=16, chunk=64):
    """Indices of the blocks on which the oracle's curve_fit raises RuntimeError."""
    jobs = [(h, tpl, blocks[s:s + chunk], s) for s in range(0, len(blocks), chunk)]
    with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
        return sorted(sum(pool.map(_raises, jobs), []))


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    w = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    from thrifty_amd import _native as F
    tpl, blocks = make(seed, h, w, nb)
    ref = reference_raises(h, tpl, blocks)
    eng = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=nb)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    gpu = np.flatnonzero(rec["flags"] & F.FLAG_FIT_UNCONVERGED).tolist()
    print("reference raises on %d blocks: %s" % (len(ref), ref))
    print("engine flags        %d blocks: %s" % (len(gpu), gpu))
    print("shared %d, reference only %s, engine only %s" % (
        len(set(ref) & set(gpu)), sorted(set(ref) - set(gpu)), sorted(set(gpu) - set(ref))))
    print(eng.path_info()["text"])


if __name__ == "__main__":
    main()
