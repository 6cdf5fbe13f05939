"""Dev/aux: throughput of other block lengths (generic multi-pass path or LDS path), device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thrifty_amd import _native as F, synth

def run(n, h, bits, sps, nblk, reps=5):
    tpl = synth.gold_template(bits, 2, sps)
    w = len(tpl); pad = h - w + 1
    win = (pad // 2, (n - w + 1) - (pad - pad // 2))
    rng = np.random.default_rng(3)
    seed, _ = synth.synth_blocks(rng, 16, n, tpl, win)
    dev = torch.device("cuda:0")
    data = torch.from_numpy(np.tile(seed, (nblk // 16, 1))).to(dev)
    out = torch.zeros(nblk * 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=nblk)
    eng.detect_device(data.data_ptr(), F.THR_IN_U8, nblk, out.data_ptr()); eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.detect_device(data.data_ptr(), F.THR_IN_U8, nblk, out.data_ptr())
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    rec = out.cpu().numpy().view(F.RECORD_DTYPE)
    print("N=%d W=%d: %.0f blocks/s (%.3f ms per %d blocks), detections %d/%d" % (
        n, w, nblk / dt, dt * 1e3, nblk, int(((rec["flags"] & 2) != 0).sum()), nblk))

if __name__ == "__main__":
    run(65536, 4096, 11, 2.0, 512)
    run(4096, 1024, 8, 2.0, 8192)
    run(16384, 4096, 10, 1.0, 8192)
