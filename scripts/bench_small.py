"""Dev/aux: throughput of the short block lengths (LDS-resident; `multipass` as argv[1]: the generic pipeline), device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thrifty_amd import _native as F, synth

PATH = sys.argv[1] if len(sys.argv) > 1 else "auto"
GEOM = {1024: (256, 7, (3, 60)), 2048: (512, 8, (5, 100)), 4096: (1024, 9, (7, 110)), 8192: (2048, 10, (7, 110))}

def run(n, nblk, reps=8):
    h, bits, cwin = GEOM[n]
    tpl = synth.gold_template(bits, 2)
    w = len(tpl); pad = h - w + 1
    win = (pad // 2, (n - w + 1) - (pad - pad // 2))
    rng = np.random.default_rng(3)
    seed, _ = synth.synth_blocks(rng, 64, n, tpl, win, carrier_bins=(cwin[0] + 3.0, min(cwin[1] - 5.0, n / 8.0)))
    dev = torch.device("cuda:0")
    data = torch.from_numpy(np.tile(seed, (nblk // 64, 1))).to(dev)
    out = torch.zeros(nblk * 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng = F.Engine(n, h, tpl, (0, 15, 0), cwin, (0, 15, 0), max_batch=nblk, path=PATH)
    eng.detect_device(data.data_ptr(), F.THR_IN_U8, nblk, out.data_ptr()); eng.sync()
    eng.profile_enable(1); eng.profile_read()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.detect_device(data.data_ptr(), F.THR_IN_U8, nblk, out.data_ptr())
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in eng.profile_read().items() if v[1]}
    rec = out.cpu().numpy().view(F.RECORD_DTYPE)
    print("N=%5d W=%4d %s: %10.0f blocks/s = %6.1f GS/s (%.3f ms per %d blocks) %s detections %d/%d" % (
        n, w, "generic" if PATH == "multipass" else "LDS    ", nblk / dt, nblk / dt * n / 1e9,
        dt * 1e3, nblk, prof, int(((rec["flags"] & 2) != 0).sum()), nblk))

if __name__ == "__main__":
    for n in (1024, 2048, 4096, 8192):
        run(n, (1 << 29) // (2 * n) // 64 * 64)     # 512 MiB of samples per launch
