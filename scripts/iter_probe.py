"""Dev/aux: where the time of `for detected, result in Detector(settings, CardStream(f))` goes.
python scripts/iter_probe.py [n_blocks] [raw]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from thrifty_amd import block_data, synth
from thrifty_amd.detect import Detector, DetectorSettings, unique_window

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
RAW = len(sys.argv) > 2

def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1.npz"), allow_pickle=False)
    n, h, tpl = int(g["block_len"]), int(g["history_len"]), g["template"]
    cthr, cwin, xthr = tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]), tuple(g["corr_thresh"])
    rng = np.random.default_rng(1)
    ook = (tpl - tpl.min()) / (tpl.max() - tpl.min()) * 2 - 1
    seed_blocks, _ = synth.synth_blocks(rng, 64, n, ook, unique_window(n, h, len(tpl)))
    st = DetectorSettings(n, h, len(tpl), cthr, cwin, tpl, xthr)
    with tempfile.TemporaryDirectory() as tmpd:
        path = os.path.join(tmpd, "rx.bin" if RAW else "rx.card")
        with open(path, "wb") as f:
            if RAW:
                step = 2 * (n - h)
                chunk = np.concatenate([seed_blocks[j][-step:] for j in range(64)]).tobytes()
                for _ in range(NB // 64):
                    f.write(chunk)
            else:
                payload = [block_data.card_line(0.0, 0, seed_blocks[j]).split(" ", 2)[2] for j in range(64)]
                for lo in range(0, NB, 4096):
                    f.write("".join("%.6f %d %s" % (1000.0 + 0.005 * i, i, payload[i % 64]) for i in range(lo, lo + 4096)).encode())
            f.flush(); os.fsync(f.fileno())
        reader = (lambda f: block_data.RawStream(f, n, h)) if RAW else (lambda f: block_data.CardStream(f, n))

        def loop(profile=False, **kw):
            with open(path, "rb") as f:
                t0 = time.perf_counter()
                det = Detector(st, reader(f), rxid=0, **kw)
                k = hits = 0
                for detected, result in det:
                    k += 1
                    hits += detected
                dt = time.perf_counter() - t0
                det.close()
            return k, hits, dt

        def loop_toad():
            with open(path, "rb") as f, open(os.path.join(tmpd, "o.toad"), "wb") as out:
                t0 = time.perf_counter()
                det = Detector(st, reader(f), rxid=0)
                det.write_toad(out)
                dt = time.perf_counter() - t0
                det.close()
            return dt

        def loop_records():
            with open(path, "rb") as f:
                t0 = time.perf_counter()
                det = Detector(st, reader(f), rxid=0)
                k = sum(len(r) for _, r in det.iter_detected_records())
                dt = time.perf_counter() - t0
                det.close()
            return k, dt
        print("write_toad: %.0f blocks/s" % (NB / loop_toad()))
        print("write_toad: %.0f blocks/s" % (NB / loop_toad()))
        for _ in range(2):
            k, hits, dt = loop()
            print("iterate: %d blocks, %d hits, %.3f s = %.0f blocks/s" % (k, hits, dt, k / dt))
        for _ in range(2):
            k, dt = loop_records()
            print("iter_detected_records: %d records %.3f s = %.0f blocks/s" % (k, dt, NB / dt))
        for kw in (dict(pin_input=False), dict(populate_threads=1), dict(populate_threads=2)):
            for _ in range(2):
                k, hits, dt = loop(**kw)
                print("iterate %s: %.3f s = %.0f blocks/s" % (kw, dt, k / dt))
        # the builder alone, in this process, while nothing else runs
        from thrifty_amd import _native as F
        recs = np.zeros(2048, dtype=F.RECORD_DTYPE); recs["flags"] = 3
        det = Detector(st, None); stamps = [0.0] * 2048
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(64):
                out = det._results(stamps, recs["block_idx"], recs)
            print("builder alone: %.0f ns per block" % ((time.perf_counter() - t0) / 64 / 2048 * 1e9))
        det.close()
        t0 = time.perf_counter(); c0 = time.process_time()
        k, hits, dt = loop()
        print("iterate: wall %.3f s, process CPU %.3f s" % (time.perf_counter() - t0, time.process_time() - c0))
        pr = cProfile.Profile()
        pr.enable()
        loop()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)

main()
