"""Dev: where thr_run_card / thr_run_stream spend their time on a file in the page cache -- the
calling thread (framing, submit, waiting), the text thread, and the input window's threads
(populate, lock, unlock, and how long the caller waited for locked segments) -- for a sweep of
populator counts, segment sizes and batch sizes.

    python scripts/run_file_probe.py [n_blocks] [card|raw]
"""
import mmap
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import _native as F
from thrifty_amd import block_data, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "card"
if len(sys.argv) > 3 and sys.argv[3] == "torch":      # the process bench.py's legs run in: torch owns streams too
    import torch
    x = torch.zeros(1 << 20, device="cuda")
    torch.cuda.synchronize()
    print("torch %s initialised on %s" % (torch.__version__, torch.cuda.get_device_name(0)))
g = np.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
n, h, tpl = int(g["block_len"]), int(g["history_len"]), g["template"]
ook = (tpl - tpl.min()) / (tpl.max() - tpl.min()) * 2 - 1
pad = h - len(tpl) + 1
rng = np.random.default_rng(0)
seed, _ = synth.synth_blocks(rng, 64, n, ook, (pad // 2, n - len(tpl) + 1 - (pad - pad // 2)))
def make_file(count):
    tmp = tempfile.NamedTemporaryFile(suffix="." + kind, delete=False)
    if kind == "card":
        lines = [block_data.card_line(0.0, 0, seed[j]).split(" ", 2)[2] for j in range(64)]
        for s0 in range(0, count, 4096):
            tmp.write("".join("%.6f %d %s" % (1000.0 + 0.005 * i, i, lines[i % 64])
                              for i in range(s0, min(count, s0 + 4096))).encode())
    else:
        step = 2 * (n - h)
        chunk = np.concatenate([seed[j][-step:] for j in range(64)]).tobytes()
        for _ in range(count // 64):
            tmp.write(chunk)
    tmp.flush()
    os.fsync(tmp.fileno())
    tmp.close()
    print("%s file %s: %d blocks, %.2f GB" % (kind, tmp.name, count, os.path.getsize(tmp.name) / 1e9))
    return tmp.name


file_a, file_b = make_file(nb), make_file(2 * nb)
thr = tuple(float(v) for v in g["carrier_thresh"])
win = tuple(int(v) for v in g["carrier_window"])
out_fd = os.open(os.devnull, os.O_WRONLY)


def run(batch, pop, seg, window=True, path=None, sleepy=False):
    path = path or file_a
    size = os.path.getsize(path)
    t_c = time.perf_counter()
    eng = F.Engine(n, h, tpl, thr, win, tuple(float(v) for v in g["corr_thresh"]), max_batch=batch)
    t_c = time.perf_counter() - t_c
    if sleepy:
        eng.set_wait_mode(True)
        print("sleeping waits; ", end="")
    cpu0 = time.process_time()
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        view = memoryview(mm)
        t0 = time.perf_counter()
        if window:
            eng.input_window(view, populate_threads=pop, segment_bytes=seg)
        if kind == "card":
            st = eng.run_card(view, out_fd=out_fd, rxid=0, batch_blocks=batch)
        else:
            st = eng.run_stream(view, first_block_idx=0, out_fd=out_fd, rxid=0, batch_blocks=batch)
        dt = time.perf_counter() - t0
        print("process CPU %.0f ms of %.0f ms wall; " % ((time.process_time() - cpu0) * 1e3, dt * 1e3), end="")
        wt = eng.debug_window_times() if window else {}
        pt = eng.debug_pipe_times()
        eng.input_window(None)
        view.release()
        mm.close()
    eng.close()
    per = 1e3 / max(1, st["batches"])
    print("create %.1f ms; " % (t_c * 1e3), end="")
    print("batch %5d pop %d seg %4d MiB%s: %.3f M blocks/s (%.1f GB/s) | per batch ms: frame %.2f submit %.2f wait %.2f "
          "| text thread: format %.2f write %.2f | window s: populate %.3f lock %.3f unlock %.3f caller-waited %.3f "
          "(%d waits, %d pageable) | submit phases ms/batch: grow %.3f h2d %.3f meta %.3f launch %.3f d2h %.3f event %.3f fill %.3f" % (
              batch, pop, seg >> 20, "" if window else " NO WINDOW", st["blocks"] / dt / 1e6, size / dt / 1e9,
              st["frame_s"] * per, st["submit_s"] * per, st["wait_s"] * per, st["format_s"] * per, st["write_s"] * per,
              wt.get("populate_s", 0), wt.get("register_s", 0), wt.get("unregister_s", 0), wt.get("acquire_wait_s", 0),
              wt.get("acquire_waits", 0), wt.get("pageable_copies", 0),
              *[pt[k] * 1e3 / max(1, pt["chunks"]) for k in ("grow_s", "h2d_s", "meta_s", "launch_s", "d2h_s", "event_s", "fill_s")]))
    print("      longest single call ms: " + " ".join("%s %.2f" % (k[4:-2], pt[k] * 1e3) for k in sorted(pt) if k.startswith("max_")))


print("-- A, A, then B (twice as long, never read before), B, A")
run(2048, 3, 128 << 20)
run(2048, 3, 128 << 20)
run(2048, 3, 128 << 20, path=file_b)
run(2048, 3, 128 << 20, path=file_b)
run(2048, 3, 128 << 20)
for pop in (1, 2):
    run(2048, pop, 128 << 20)
run(2048, 1, 128 << 20, sleepy=True)       # a rank of eight on a 16-CPU host
for seg in (32 << 20, 256 << 20):
    run(2048, 3, seg)
run(4096, 3, 128 << 20)
run(2048, 3, 128 << 20, window=False)
os.unlink(file_a)
os.unlink(file_b)
