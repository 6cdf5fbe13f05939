"""Dev: where the host thread spends its time on the `.card` file -> `.toad` text path (per engine
batch: framing, submit, collect wait, text formatting)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import _native, block_data, synth
from thrifty_amd.detect import Detector, DetectorSettings

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pin = (sys.argv[2] != "nopin") if len(sys.argv) > 2 else True
rng = np.random.default_rng(0)
if len(sys.argv) > 3 and sys.argv[3] == "c1":      # bench.py's card_to_toad leg: the example detector.cfg
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c1.npz"))
    n, h, tpl = int(g["block_len"]), int(g["history_len"]), g["template"]
    ook = (tpl - tpl.min()) / (tpl.max() - tpl.min()) * 2 - 1
    pad = h - len(tpl) + 1
    seed, _ = synth.synth_blocks(rng, 32, n, ook, (pad // 2, n - len(tpl) + 1 - (pad - pad // 2)))
    st = DetectorSettings(n, h, len(tpl), tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]), tpl, tuple(g["corr_thresh"]))
else:
    n, h = 16384, 4096
    tpl = synth.gold_template(10, 2)
    seed, _ = synth.synth_blocks(rng, 32, n, tpl, (1537, 13825))
    st = DetectorSettings(n, h, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
line = [block_data.card_line(1000.0 + i, i, seed[i % 32]) for i in range(32)]
text = "".join(line[i % 32] for i in range(nb)).encode()
tmp = tempfile.NamedTemporaryFile(suffix=".card", delete=False)
tmp.write(text); tmp.close()
for rep in range(3):
    f = open(tmp.name, "rb")
    t_open = time.perf_counter()
    det = Detector(st, block_data.CardStream(f, n), rxid=0, pin_input=pin)
    t_open = time.perf_counter() - t_open
    eng, cs = det._engine, det._card
    T = dict(frame=0.0, submit=0.0, collect=0.0, fmt=0.0)
    t_all = time.perf_counter()
    pending, nout, nbatch = None, 0, 0
    while True:
        t0 = time.perf_counter()
        b = cs.next_batch(det.batch_size)
        t1 = time.perf_counter()
        T["frame"] += t1 - t0
        cur = None
        if b is not None:
            stamps, idxs, txt, offs = b
            cur = (stamps, idxs, eng.submit_card(txt, offs, idxs))
            T["submit"] += time.perf_counter() - t1
            nbatch += 1
        if pending is not None:
            t2 = time.perf_counter()
            recs = eng.collect(pending[2])[:, 0]
            t3 = time.perf_counter()
            keep = np.flatnonzero(recs["flags"] & _native.FLAG_CORR)
            out = _native.format_toad(recs[keep], np.asarray(pending[0], dtype=np.float64)[keep], det.new_len, rxid=0)
            nout += len(out)
            T["collect"] += t3 - t2
            T["fmt"] += time.perf_counter() - t3
        pending = cur
        if cur is None:
            break
    dt = time.perf_counter() - t_all
    print("pin=%s %d blocks %d batches of %d: %.0f blocks/s (%.0f with the %.1f ms of Detector()); per batch ms: %s; text out %d B" % (
        pin, nb, nbatch, det.batch_size, nb / dt, nb / (dt + t_open), t_open * 1e3,
        {k: round(v / nbatch * 1e3, 3) for k, v in T.items()}, nout))
    f.close()
os.unlink(tmp.name)
# the product loop itself (Detector.iter_toad_text: two batches ahead on a page-locked file)
tmp2 = tempfile.NamedTemporaryFile(suffix=".card", delete=False)
tmp2.write(text); tmp2.close()
for rep in range(3):
    with open(tmp2.name, "rb") as f:
        t0 = time.perf_counter()
        det = Detector(st, block_data.CardStream(f, n), rxid=0, pin_input=pin)
        out = b"".join(det.iter_toad_text())
        dt = time.perf_counter() - t0
    print("Detector.iter_toad_text, whole job: %.0f blocks/s (depth %d, %d B of text)" % (nb / dt, det._depth, len(out)))
    del det
os.unlink(tmp2.name)
