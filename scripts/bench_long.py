"""Dev/aux: per-kernel times of the long-block path (N = 65536)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thrifty_amd import _native as F, synth

n, h = 65536, 4096
tpl = synth.gold_template(11, 2, 2.0)
w = len(tpl); pad = h - w + 1
win = (pad // 2, (n - w + 1) - (pad - pad // 2))
rng = np.random.default_rng(3)
seed, _ = synth.synth_blocks(rng, 16, n, tpl, win)
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
data = torch.from_numpy(np.tile(seed, (nblk // 16, 1))).to(dev)
out = torch.zeros(nblk * 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=nblk)
eng.detect_device(data.data_ptr(), F.THR_IN_U8, nblk, out.data_ptr()); eng.sync()
eng.profile_enable(1); eng.profile_read()
t0 = time.perf_counter()
for _ in range(4):
    eng.detect_device(data.data_ptr(), F.THR_IN_U8, nblk, out.data_ptr())
eng.sync()
dt = (time.perf_counter() - t0) / 4
prof = eng.profile_read()
print("N=65536: %.0f blocks/s; per %d blocks: %s" % (nblk / dt, nblk, {k: round(v[0] / 4, 3) for k, v in prof.items()}))
