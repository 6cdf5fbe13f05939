"""Dev/aux: host-buffer entry point (thr_detect from pageable NumPy memory): blocks/s incl. PCIe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import _native as F, synth

n, h = 16384, 4096
tpl = synth.gold_template(10, 2)
rng = np.random.default_rng(0)
seed, _ = synth.synth_blocks(rng, 64, n, tpl, (1537, 13825))
for batch in (1024, 4096):
    nb = 16 * 4096
    blocks = np.tile(seed, (nb // 64, 1))          # 2 GiB of pageable u8
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=batch)
    eng.detect(blocks[:batch])
    t0 = time.perf_counter()
    rec = eng.detect(blocks)
    dt = time.perf_counter() - t0
    print("thr_detect, pageable host memory, max_batch %d: %.0f blocks/s = %.1f GB/s (%d detections of %d)" % (
        batch, nb / dt, nb * 2 * n / dt / 1e9, int(((rec["flags"] & 2) != 0).sum()), nb))
