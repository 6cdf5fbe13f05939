"""Dev/aux: `thrifty detect rx.card --quiet -o rx.toad` end to end (in-process detector_cli call):
text file -> framing -> device decode + detection -> .toad text.  Dense = every block detects."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import block_data, synth
from thrifty_amd.detect import Detector, detector_cli

n, h = 16384, 4096
tpl = synth.gold_template(10, 2)
rng = np.random.default_rng(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = tempfile.mkdtemp()
np.save(os.path.join(d, "template.npy"), tpl.astype(np.float64))
open(os.path.join(d, "detector.cfg"), "w").write(
    "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\ncarrier_window: 7 - 110\n"
    "carrier_threshold: 15 * snr\ncorr_threshold: 15*snr\ntemplate: %s\n" % os.path.join(d, "template.npy"))
for name, frac in (("dense", 1.0), ("sparse (10 % signal)", 0.1)):
    seed, _ = synth.synth_blocks(rng, 64, n, tpl, (1537, 13825), signal_frac=frac)
    path = os.path.join(d, "rx.card")
    with open(path, "w") as f:
        for i in range(nb):
            f.write(block_data.card_line(1000.0 + 0.005 * i, i, seed[i % 64]))
    for rep in range(2):     # (first pass also pays library load and page-cache fill)
        t0 = time.perf_counter()
        detector_cli(Detector, argv=[path, "--quiet", "-o", os.path.join(d, "rx.toad"), "-c", os.path.join(d, "detector.cfg")])
        dt = time.perf_counter() - t0
    lines = sum(1 for _ in open(os.path.join(d, "rx.toad")))
    print("%-22s %d blocks, %.1f MB of text: %.0f blocks/s (%d detections written)" % (
        name, nb, os.path.getsize(path) / 1e6, nb / dt, lines))
