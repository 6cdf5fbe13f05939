"""Dev aid: print the per-wave cycle timeline of one k_correlate block (needs a library
built with THR_EXTRA_CFLAGS="-DTHR_DEV -DTHR_DEV_MINIMAL")."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from thrifty_amd import _native as F, synth
if os.environ.get("THR_DEV_ABI"): F.ABI_VERSION = int(os.environ["THR_DEV_ABI"])   # (a dev library of an older header)

NW = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PATH = sys.argv[2] if len(sys.argv) > 2 else "auto"
n, h = 16384, 4096
tpl = synth.gold_template(10, 2)
rng = np.random.default_rng(1)
win = (1537, 13825)
B = 32768
blocks, _ = synth.synth_blocks(rng, 256, n, tpl, win)
eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=B, path=PATH)
dev = torch.device("cuda:0")
data = torch.from_numpy(np.tile(blocks, (B // 256, 1))).to(dev)
out = torch.zeros(B * 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    eng.detect_device(data.data_ptr(), F.THR_IN_U8, B, out.data_ptr())
eng.sync()
buf = (C.c_uint64 * 128)()
eng._lib.thr_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
eng._lib.thr_debug_timeline(eng._h, buf)
tl = np.array(buf[:], dtype=np.uint64).reshape(8, 16)[:NW].astype(np.int64)
names = ["top", "nxt issued", "P1 done", "phasor nxt", "barrier1", "P2", "P3", "mult+PA", "PB",
         "barrier2", "PC", "stats", "reduce", "end"]
t0 = tl[:, 0].min()
print("stamp (cycles since earliest top), per wave; then per-phase deltas for wave 0..7")
for i, nm in enumerate(names):
    print("%-12s" % nm, " ".join("%7d" % (tl[w, i] - t0) for w in range(NW)))
print()
for i in range(1, len(names)):
    print("d %-10s" % names[i], " ".join("%7d" % (tl[w, i] - tl[w, i - 1]) for w in range(NW)))
