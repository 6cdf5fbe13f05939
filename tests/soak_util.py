"""Shared by tests/test_gpu_soak.py and tests/tools/soak_parity.py: the CPU oracle spread over
worker processes (spawned -- the parent holds a HIP context), and the record comparison."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def oracle_rows(args):
    """(worker) -> (lo, [(cbin, cdet, coff, cenergy, sample, det, energy, offset, noise), ...])"""
    os.environ["OMP_NUM_THREADS"] = "1"
    lo, blocks, n, h, tpl, cthr, cwin, xthr = args
    from oracle import thrifty_np as onp
    orc = onp.OracleDetector(n, h, tpl, cthr, cwin, xthr)
    out = []
    for i in range(len(blocks)):
        try:
            (r,) = orc.detect_u8(lo + i, blocks[i])
        except IndexError:      # the reference raises when peak_idx + 3 >= N (carrier_sync.py:187)
            out.append(None)
            continue
        c = r.corr
        out.append((r.carrier.bin, bool(r.carrier.detected), float(r.carrier.offset), float(r.carrier.energy),
                    int(c.sample) if c else -1, bool(c.detected) if c else False,
                    float(c.energy) if c else 0.0, float(c.offset) if c else 0.0,
                    float(c.noise) if c else 0.0))
    return lo, out


def run_oracle(blocks, n, h, tpl, cthr, cwin, xthr, procs=None, chunk=64):
    procs = procs or max(1, min(32, (os.cpu_count() or 2) // 2))
    jobs = [(s, blocks[s:s + chunk], n, h, tpl, cthr, cwin, xthr) for s in range(0, len(blocks), chunk)]
    rows = [None] * len(blocks)
    # (an executor, not mp.Pool: a worker that dies raises BrokenProcessPool instead of being
    # respawned forever, and every result has a deadline)
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(min(procs, len(jobs)), mp_context=mp.get_context("spawn")) as pool:
        for lo, out in pool.map(oracle_rows, jobs, timeout=900):
            rows[lo:lo + len(out)] = out
    return rows


def compare(rec, rows, blocks, flag_carrier=1, flag_corr=2, flag_index_error=4, only=None):
    """-> (mismatch counts, worst deviations, [indices of carrier-bin ties]).  Exact fields (bin,
    sample, verdicts) are counted over every block; the worst deviations only over the blocks
    where `only` (bool array) is set, when given."""
    from oracle import thrifty_np as onp
    mism = dict(bin=0, carrier=0, sample=0, det=0, index_error=0)
    worst = dict(energy=0.0, offset=0.0, car_off=0.0, car_energy=0.0, noise=0.0)
    ties = []
    for i, row in enumerate(rows):
        r = rec[i]
        if row is None or (r["flags"] & flag_index_error):
            mism["index_error"] += (row is None) != bool(r["flags"] & flag_index_error)
            continue
        cbin, cdet, coff, cen, samp, det, en, off, noise = row
        if r["carrier_bin"] != cbin:
            # inherent tie: the two bins' float32 magnitudes are equal (to an ulp) in NumPy itself and
            # the two FFT implementations round differently -- counted apart, never silently
            mag = np.abs(np.fft.fft(onp.iq_u8_to_c64(blocks[i])))
            a, b = np.float32(mag[int(r["carrier_bin"]) % len(mag)]), np.float32(mag[cbin % len(mag)])
            if abs(a - b) <= np.spacing(max(a, b)):
                ties.append(i)
                continue
            mism["bin"] += 1
            continue
        mism["carrier"] += bool(r["flags"] & flag_carrier) != cdet
        if not cdet or bool(r["flags"] & flag_carrier) != cdet:
            continue
        mism["sample"] += r["corr_sample"] != samp
        mism["det"] += bool(r["flags"] & flag_corr) != det
        if only is not None and not only[i]:
            continue
        worst["car_energy"] = max(worst["car_energy"], abs(r["carrier_energy"] - cen) / abs(cen))
        worst["car_off"] = max(worst["car_off"], abs(r["carrier_offset"] - coff))
        if r["corr_sample"] == samp:
            worst["energy"] = max(worst["energy"], abs(r["corr_energy"] - en) / abs(en))
            worst["noise"] = max(worst["noise"], abs(r["corr_noise"] - noise) / abs(noise))
            if det and bool(r["flags"] & flag_corr):
                worst["offset"] = max(worst["offset"], abs(r["corr_offset"] - off))
    return mism, worst, ties
