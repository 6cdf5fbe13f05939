"""Dev/aux: end-to-end `.card text -> records` rate: host decode (card_reader) vs device decode (CardStream)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import block_data, synth
from thrifty_amd.detect import Detector, DetectorSettings

n, h = 16384, 4096
tpl = synth.gold_template(10, 2)
rng = np.random.default_rng(0)
seed, _ = synth.synth_blocks(rng, 32, n, tpl, (1537, 13825))
nb = 4096
text = "".join(block_data.card_line(1000.0 + i, i, seed[i % 32]) for i in range(nb)).encode()
st = DetectorSettings(n, h, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
import tempfile
tmp = tempfile.NamedTemporaryFile(suffix=".card", delete=False)
tmp.write(text); tmp.close()
for name, mk in (("host decode (card_reader)", lambda: block_data.card_reader(io.StringIO(text.decode()))),
                 ("device decode (CardStream, pipe)", lambda: block_data.CardStream(io.BytesIO(text), n)),
                 ("device decode (CardStream, file)", lambda: block_data.CardStream(open(tmp.name, "rb"), n))):
    det = Detector(st, mk())
    t0 = time.perf_counter()
    cnt = sum(1 for d, r in det if d)
    dt = time.perf_counter() - t0
    print("%-34s %8.0f blocks/s (%d detections, %.1f MB of text)" % (name, nb / dt, cnt, len(text) / 1e6))
