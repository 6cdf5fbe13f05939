"""Dev/aux: end-to-end `raw u8 stream -> records` rate: host framing (block_reader) vs
device framing (RawStream -> thr_detect_stream)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import block_data, synth
from thrifty_amd.detect import Detector, DetectorSettings

n, h = 16384, 4096
new = n - h
tpl = synth.gold_template(10, 2)
rng = np.random.default_rng(0)
nb = 8192
x = (rng.normal(0, 0.02, new * 64) + 1j * rng.normal(0, 0.02, new * 64))
ook = 0.3 * (np.asarray(tpl, float) + 1) / 2
for j in range(64):
    s = j * new + 3000 + 97 * j
    k = np.arange(len(tpl))
    x[s:s + len(tpl)] += ook * np.exp(2j * np.pi * (20 + j) * (k + s) / n)
raw = np.tile(synth.quantise_iq(x), nb // 64).tobytes()
st = DetectorSettings(n, h, len(tpl), (0, 15, 0), (7, 110), tpl, (0, 15, 0))
import tempfile
tmp = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
tmp.write(raw); tmp.close()
for name, mk in (("host framing (block_reader)", lambda: block_data.block_reader(io.BytesIO(raw), n, h)),
                 ("device framing (RawStream, pipe)", lambda: block_data.RawStream(io.BytesIO(raw), n, h)),
                 ("device framing (RawStream, file)", lambda: block_data.RawStream(open(tmp.name, "rb"), n, h))):
    det = Detector(st, mk())
    t0 = time.perf_counter()
    cnt = sum(1 for d, r in det if d)
    dt = time.perf_counter() - t0
    print("%-30s %8.0f blocks/s (%d detections, %.1f MB stream, %.1f MS/s)" % (
        name, nb / dt, cnt, len(raw) / 1e6, nb * new / dt / 1e6))

# sparse stream (a burst in every 10th block), detections only -- what `thrifty detect --raw --quiet` does
x2 = (rng.normal(0, 0.02, new * 640) + 1j * rng.normal(0, 0.02, new * 640))
for j in range(64):
    s = 10 * j * new + 3000 + 97 * j
    k = np.arange(len(tpl))
    x2[s:s + len(tpl)] += ook * np.exp(2j * np.pi * (20 + j) * (k + s) / n)
nb2 = 640 * 32
raw2 = np.tile(synth.quantise_iq(x2), nb2 // 640).tobytes()
tmp2 = tempfile.NamedTemporaryFile(suffix=".bin", delete=False)
tmp2.write(raw2); tmp2.close()
det = Detector(st, block_data.RawStream(open(tmp2.name, "rb"), n, h), batch_size=2048)
det.only_detections = True
t0 = time.perf_counter()
cnt = sum(1 for d, r in det if d)
dt = time.perf_counter() - t0
print("%-30s %8.0f blocks/s (%d detections, %.1f MB stream, %.1f MS/s)" % (
    "sparse file, detections only", nb2 / dt, cnt, len(raw2) / 1e6, nb2 * new / dt / 1e6))

# the CLI's quiet path (`thrifty detect --raw --quiet -o`): no per-block Python objects, text from thr_format_toad
for label, path, nblk in (("dense file -> .toad text", tmp.name, nb), ("sparse file -> .toad text", tmp2.name, nb2)):
    det = Detector(st, block_data.RawStream(open(path, "rb"), n, h), rxid=0)
    t0 = time.perf_counter()
    nbytes = sum(len(tx) for tx in det.iter_toad_text())
    dt = time.perf_counter() - t0
    print("%-30s %8.0f blocks/s (%.1f MB of .toad text, %.1f MS/s)" % (label, nblk / dt, nbytes / 1e6, nblk * new / dt / 1e6))
