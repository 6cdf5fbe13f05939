"""Dev/aux: `identify` (classification + duplicate filter + output order) on n detections:
thr_identify on the GPU (host columns in, host columns out) vs the NumPy oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import thrifty_np as onp
from thrifty_amd import _native as F

for n in (10_000, 1_000_000, 8_000_000):
    rng = np.random.default_rng(n)
    centres = np.array([30, 61, 93])
    rxid = rng.integers(0, 4, n).astype(np.int32)
    cbin = (centres[rng.integers(0, 3, n)] + rxid + np.round(rng.normal(0, 0.7, n))).astype(np.int32)
    coff = rng.uniform(-0.5, 0.5, n)
    block = rng.integers(0, n // 3, n).astype(np.int32)
    ts = 1.7e9 + block * 0.00512 + rng.uniform(0, 1e-3, n)
    energy = rng.uniform(50, 200, n)
    F.identify(rxid[:10], block[:10], ts[:10], cbin[:10], coff[:10], energy[:10])   # warm up
    t0 = time.perf_counter()
    txid, keep, order = F.identify(rxid, block, ts, cbin, coff, energy)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    want_tx, _ = onp.auto_classify(rxid, cbin)
    want_keep = onp.duplicate_mask(rxid, want_tx, block, ts, energy)
    want_order = onp.filter_order(want_keep, ts)
    t_cpu = time.perf_counter() - t0
    ok = np.array_equal(txid, want_tx) and np.array_equal(keep, want_keep) and np.array_equal(order, want_order)
    print("n=%9d  gpu %.4f s (%.1f M det/s, PCIe + allocations included)  numpy %.3f s (%.2f M det/s)  equal=%s"
          % (n, t_gpu, n / t_gpu / 1e6, t_cpu, n / t_cpu / 1e6, ok))
