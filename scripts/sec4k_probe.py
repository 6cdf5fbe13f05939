"""Dev/aux: block_len 16384 with a short template -- the sectioned correlate stage (detect16k_sec.hip,
4096-sample sections, the default) against the unsectioned kernel (path="unsectioned") on the same
device-resident blocks: records compared field by field, per-kernel times from the engines' own HIP
events.    python scripts/sec4k_probe.py [n_blocks] [history] [template_len] [n_templates]
(n_templates > 1: Gold codes 2, 3, ... as bench.py's t4 leg; blocks carry template 0's burst)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thrifty_amd import _native as F, synth

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1023
NTPL = int(sys.argv[4]) if len(sys.argv) > 4 else 1
N = 16384

def main():
    rng = np.random.default_rng(11)
    tpl = synth.gold_template(10, 2) if W == 1023 else np.sign(rng.normal(0, 1, W))
    pad = H - W + 1
    win = (pad // 2, (N - W + 1) - (pad - pad // 2))
    seed, _ = synth.synth_blocks(rng, 256, N, tpl, win, carrier_bins=(10.0, 100.0))
    tpls = tpl if NTPL == 1 else np.stack([synth.gold_template(10, 2 + i) for i in range(NTPL)]).astype(np.float64)
    dev = torch.device("cuda:0")
    data = torch.from_numpy(np.tile(seed, (NB // 256, 1))).to(dev)
    thr = (0, 15, 0)
    recs, rates = {}, {}
    paths = ("auto", "auto") if os.environ.get("SEC_ONLY") else ("unsectioned", "auto", "unsectioned", "auto")
    for path in paths:
        out = torch.zeros(NB * 64 * NTPL, dtype=torch.uint8, device=dev)
        eng = F.Engine(N, H, tpls, thr, (7, 110), thr, max_batch=NB, path=path)
        eng.detect_device(data.data_ptr(), F.THR_IN_U8, NB, out.data_ptr()); eng.sync()
        eng.profile_enable(1); eng.profile_read()
        reps = 12
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.detect_device(data.data_ptr(), F.THR_IN_U8, NB, out.data_ptr())
        eng.sync()
        dt = (time.perf_counter() - t0) / reps
        prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in eng.profile_read().items() if v[1]}
        recs[path] = out.cpu().numpy().view(F.RECORD_DTYPE).copy()
        print("%-12s sections %s: %.3f ms per %d blocks = %.2f M blocks/s  %s" % (
            path, eng.sections() if hasattr(eng, "sections") else "?", dt * 1e3, NB, NB / dt / 1e6, prof), flush=True)
        eng.close()
    if os.environ.get("SEC_ONLY"):
        return
    a, b = recs["unsectioned"], recs["auto"]
    det = (a["flags"] & F.FLAG_CORR) != 0
    print("detections %d / %d (sectioned %d)" % (det.sum(), NB, ((b["flags"] & F.FLAG_CORR) != 0).sum()))
    for f in ("flags", "carrier_bin", "corr_sample", "block_idx"):
        print("  %-14s differing: %d" % (f, int((a[f] != b[f]).sum())))
    for f in ("corr_offset", "corr_energy", "corr_noise", "carrier_offset", "carrier_energy"):
        x, y = a[f][det].astype(np.float64), b[f][det].astype(np.float64)
        rel = np.abs(x - y) / np.maximum(np.abs(x), 1e-30) if f != "corr_offset" else np.abs(x - y)
        print("  %-14s max %s difference: %.3g" % (f, "abs" if f == "corr_offset" else "rel", rel.max() if len(rel) else 0.0))
    bad = np.nonzero(a["corr_sample"] != b["corr_sample"])[0][:8]
    for i in bad:
        print("   block %d: unsectioned %s / sectioned %s" % (i, a[i], b[i]))

if __name__ == "__main__":
    main()
