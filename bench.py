#!/usr/bin/env python3
"""Benchmark of the `thrifty detect` hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c1] [--templates T]
                    [--batch B] [--mix dense|sparse] [--scaling weak|strong]

One "step" = one pass of the hot path (FFT -> carrier detect -> fit -> shift ->
FFT -> x conj(T) -> IFFT -> SoA) over one batch of synthetic IQ blocks that are
already resident in HBM.  A step's batch is R launch batches of B blocks (B = the
engine's max_batch, 32768 at N = 16384); R is chosen after a calibration burst so
that the K timed steps last at least --min-seconds (default 2 s: a sustained rate
on a part that clocks to its power budget, not a 40 ms burst).  The default run
then adds bounded legs for the other single-GPU configs (`configs`: c3, t4, sparse,
fullwin, c3t4, c1, c1_sparse), each with its own recomputable roofline figures, the CPU
baseline, and the file legs (.card / raw file -> .toad).  Workloads (BASELINE.json `configs`):

  --config c2 (default, the headline): configs[1] -- block_len 16384, history 4096,
      1023-chip Gold template, K*B (default 32 * 32768 = 1 Mi) blocks per GPU
  --config c2 --templates 4: configs[4] in its 1-GPU form -- 4 TX Gold templates per block
  --config c3: configs[2] -- block_len 65536, history 4096, 2047-chip Gold code at 2 samples
      per chip (W = 4094), the long-FFT regime (a block does not fit the LDS)
  --config c1: configs[0]'s geometry kernel-resident -- the example detector.cfg (history 4920,
      the 4914-sample extracted template; tests/golden/c1.npz holds both)

Output: ONE JSON line on stdout, < 7 KB (the driver keeps the last 8 KB): the contract keys,
`roofline` and `cpu_baseline` as flat objects, `configs` with the agreed compact keys per leg, and
LAST a flat `summary` of every leg's blocks/s.  Everything else (every kernel's mean duration per
leg, the VALU view, workload prose, the file loops' time budget) goes to
gpurun_out/bench_detail_<config>_n<N>.json and to stderr.

Metric: IQ blocks/s, whole job.  For N > 1 the driver launches
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
and a plain `python bench.py --gpus N` starts exactly that by itself; both routes apply the same
rank environment first (parallel.rank_env).  Blocks are sharded by contiguous block-index ranges
(weak scaling: K*B blocks per GPU; --scaling strong: one GPU's job split over the ranks); the only
collective is the gather of detection records to rank 0, rehearsed in the pre-flight.
`--dist-backend gloo` rehearses the whole N-rank body (pre-flight, gather rehearsal, step-size
broadcast, MAX all-reduce of the time, record gather) on ONE GPU: every rank computes on cuda:0
and the collectives carry CPU tensors.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260928
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X packed-fp32 VALU peak (MI355X_MICROARCH.md)
WINDOW_BINS = (7, 110)
THRESH = (0, 15, 0)

# name -> (block_len, history, Gold register bits, samples per chip, default batch, resident blocks,
#          BASELINE config index, handles per GPU)
CONFIGS = {
    "c2": dict(n=16384, h=4096, bits=10, sps=1.0, batch=32768, resident=1 << 20, idx=1, streams=2,
               label="block_len=16384 history=4096 1023-chip Gold template"),
    # (N > 16384 on ONE handle: both stages fill the chip, a second handle measured -1 %)
    "c3": dict(n=65536, h=4096, bits=11, sps=2.0, batch=16384, resident=1 << 18, idx=2, streams=1,
               label="block_len=65536 history=4096 2047-chip Gold code at 2 samples/chip (W=4094)"),
    # BASELINE configs[0]'s geometry (the reference's example/detector.cfg + example/template.npy, also
    # what rpi/detector.cfg deploys) kernel-resident: the template and settings are the data fixture
    # tests/golden/c1.npz (history 4920, 4914-sample extracted template, window bins 7..110)
    "c1": dict(n=16384, h=4920, fixture="c1.npz", batch=32768, resident=1 << 19, idx=0, streams=2,
               label="block_len=16384 history=4920 4914-sample extracted template (example/detector.cfg)"),
}


# extra legs of the default run: other single-GPU workloads of BASELINE.json's configs (and the
# reference's DEFAULT carrier window, settings.py:75-80 '0--1' = every bin, which takes the
# full-spectrum carrier kernel instead of the pruned one)
LEGS = {"c3": dict(name="c3", T=1, mix="dense", batch=16384, resident=2 * 16384),
        "t4": dict(name="c2", T=4, mix="dense", batch=32768, resident=2 * 32768),
        "sparse": dict(name="c2", T=1, mix="sparse", batch=32768, resident=4 * 32768),
        "fullwin": dict(name="c2", T=1, mix="dense", batch=32768, resident=4 * 32768, window=(0, -1)),
        "c3t4": dict(name="c3", T=4, mix="dense", batch=4096, resident=2 * 4096),
        "c1": dict(name="c1", T=1, mix="dense", batch=32768, resident=4 * 32768),
        "c1_sparse": dict(name="c1", T=1, mix="sparse", batch=32768, resident=4 * 32768)}


def csrc_sha16():
    """Hash of the kernel sources: profiles/hbm_traffic.json records the one its PMC passes ran on."""
    from thrifty_amd import build
    return build.csrc_hash()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=0,
                    help="blocks per step (per GPU); default per config (c2: 32768 x 32 steps = "
                         "BASELINE's 1 Mi blocks)")
    ap.add_argument("--mix", choices=["dense", "sparse"], default="dense",
                    help="dense: every block carries a signal; sparse: 10%% do")
    ap.add_argument("--templates", type=int, default=1)
    ap.add_argument("--carrier-window", type=int, nargs=2, default=list(WINDOW_BINS), metavar=("START", "STOP"),
                    help="carrier window bins (default 7 110, SURVEY 8(d)); 0 -1 = every bin, the "
                         "reference's own default (settings.py:75-80), runs the full-spectrum carrier kernel")
    ap.add_argument("--streams", type=int, default=0,
                    help="engine handles (each with its own HIP stream) the steps alternate over "
                         "(default per config)")
    ap.add_argument("--profile-kernels", type=int, default=16,
                    help="n > 0: HIP events around the kernels of every n-th step of the timed "
                         "region (roofline leg; 1 = every step, costs ~4%%); 0 = off")
    ap.add_argument("--variant", choices=["default", "preshift"], default="default",
                    help="default: reference Detector (the headline); preshift: the reference's "
                         "experimental PreshiftDetector (one fused kernel per block)")
    ap.add_argument("--preshift-num", type=int, default=21, help="bank size of --variant preshift")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="lower bound on the timed region: a step becomes R launch batches of --batch "
                         "blocks, R from a calibration burst, so that K steps last this long (0: R = 1)")
    ap.add_argument("--legs", default="auto",
                    help="comma list of extra single-GPU legs after the main one (%s); 'auto' = all of "
                         "them on the default c2 / 1 GPU run, none otherwise; 'none'" % ",".join(LEGS))
    ap.add_argument("--leg-seconds", type=float, default=0.75, help="timed region of each extra leg")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="worker processes of the all-cores CPU leg (-1: one per physical core, "
                         "0: skip the leg)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall budget of the single-core CPU baseline leg (0 disables every CPU leg)")
    ap.add_argument("--card-blocks", type=int, default=131072,
                    help="blocks of the config-#1 .card -> .toad plumbing leg (0 skips it)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl: RCCL, one GPU per rank (the real thing); gloo: every rank on cuda:0, "
                         "collectives on CPU tensors -- a rehearsal of the N-rank body on a 1-GPU box")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every GPU runs the full per-GPU job (K x R launch batches, the "
                         "config's resident blocks each); strong: the job one GPU would run -- BASELINE "
                         "configs[3]: 1 Mi blocks in total -- split over the N ranks")
    ap.add_argument("--resident-blocks", type=int, default=0,
                    help="distinct blocks resident in HBM per rank (default per config: c2 1 Mi = 32 GiB)")
    return ap.parse_args()


def synth_on_device(torch, dev, gen, n_blocks, n, template, window, signal_frac, chunk=0, truth=None):
    """SURVEY.md 8(d) generator, on the GPU: OOK burst 0.3*(t+1)/2 at a uniform lag in
    the unique window, carrier bin ~ U(10,100), AWGN sigma 0.02, u8 quantiser
    (x*128 + 127.4, truncating).  Returns uint8 [n_blocks, 2N]; if `truth` is a dict it
    receives the drawn lag / carrier bin / has-signal tensors (tests/test_gpu_fullsize.py)."""
    w = len(template)
    lo, hi = window
    chunk = chunk or max(1, (2048 * 16384) // n)
    out = torch.empty((n_blocks, 2 * n), dtype=torch.uint8, device=dev)
    ook = torch.as_tensor(0.3 * (template + 1) / 2, dtype=torch.float32, device=dev)
    ar = torch.arange(w, device=dev)
    for s in range(0, n_blocks, chunk):
        b = min(chunk, n_blocks - s)
        x = torch.randn((b, n, 2), generator=gen, device=dev, dtype=torch.float32) * 0.02
        pos = torch.randint(lo, hi, (b,), generator=gen, device=dev)
        car = torch.rand((b,), generator=gen, device=dev, dtype=torch.float64) * 90.0 + 10.0
        has = torch.rand((b,), generator=gen, device=dev) < signal_frac
        idx = pos[:, None] + ar[None, :]                                   # [b, w]
        ph = (2 * np.pi / n) * car[:, None] * idx.to(torch.float64)         # float64 phase
        amp = ook[None, :] * has[:, None].to(torch.float32)
        rows = torch.arange(b, device=dev)[:, None].expand(b, w)
        x[rows, idx, 0] += amp * torch.cos(ph).to(torch.float32)
        x[rows, idx, 1] += amp * torch.sin(ph).to(torch.float32)
        q = (x * 128.0 + 127.4).clamp_(0, 255).to(torch.uint8)
        out[s:s + b] = q.view(b, 2 * n)
        if truth is not None:
            truth.setdefault("pos", []).append(pos)
            truth.setdefault("car", []).append(car)
            truth.setdefault("has", []).append(has)
        del x, q
    return out


def make_oracle(n, h, template, preshift_num=0, window=WINDOW_BINS):
    from oracle import thrifty_np as onp
    if preshift_num:
        return onp.OraclePreshiftDetector(n, h, template, THRESH, tuple(window), THRESH, num=preshift_num)
    return onp.OracleDetector(n, h, template, THRESH, tuple(window), THRESH)


def cpu_baseline(n, h, blocks_u8, idx, template, budget_s, gpu_rec, preshift_num=0, window=WINDOW_BINS):
    """Oracle (oracle/thrifty_np.py, a NumPy port of the reference algorithm) timed on ONE
    host core over a bounded sample of the same blocks; also spot-checks parity."""
    from thrifty_amd import _native as F
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    orc = make_oracle(n, h, template, preshift_num, window)
    done, mism = 0, 0
    t0 = time.perf_counter()
    for i in range(len(blocks_u8)):
        res = orc.detect_u8(int(idx[i]), blocks_u8[i])
        if not preshift_num:
            (res,) = res
        r = gpu_rec[i]
        ok = (r["carrier_bin"] == res.carrier.bin and
              bool(r["flags"] & F.FLAG_CARRIER) == res.carrier.detected)
        if ok and res.carrier.detected:
            ok = (r["corr_sample"] == res.corr.sample and
                  bool(r["flags"] & F.FLAG_CORR) == res.corr.detected and
                  abs(r["corr_energy"] - res.corr.energy) <= 1e-4 * abs(res.corr.energy) and
                  abs(r["corr_offset"] - res.corr.offset) <= 1e-4 + 1e-4 * abs(res.corr.offset))
        mism += 0 if ok else 1
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 32:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "blocks/s", "cores": 1, "kind": "port",
            "sample": "%d of the benchmark's own u8 blocks (template 0) through "
                      "oracle/thrifty_np.py (NumPy %s pocketfft + SciPy curve_fit), 1 thread, %.1f s"
                      % (done, np.__version__, dt),
            "parity_checked": done, "parity_mismatches": mism}


def _compute_view(kernel, n, n_templates, blocks_per_launch, avg_ms, n_sec=0):
    """Nominal flop of the dominant kernel per block -> achieved TFLOP/s vs the VALU peak."""
    fft = 5.0 * n * np.log2(n)
    point = 6.0 * n
    if kernel == "k_correlate_4k":   # per block n_sec sections of 4096 points: shift, FFT, product, IFFT, |.|^2
        m = 4096.0
        flop = n_sec * (6.0 * m + 5.0 * m * 12 + n_templates * (6.0 * m + 5.0 * m * 12 + 3.0 * m))
    elif kernel in ("k_correlate", "k_correlate_sub", "k_correlate_seg"):  # shift, FFT#2, then per template: product, IFFT, |.|^2
        flop = point + fft + n_templates * (point + fft + 3.0 * n)
    elif kernel == "k_preshift":     # FFT#1, |X|^2, product, IFFT, |.|^2
        flop = fft + 3.0 * n + point + fft + 3.0 * n
    else:                            # carrier stage: FFT#1 (pruned variants do less) + |X|^2
        flop = fft + 3.0 * n
    tflops = flop * blocks_per_launch / (avg_ms * 1e-3) / 1e12
    return {"flop_per_block_nominal": flop, "achieved_tflops": tflops,
            "peak_tflops": FP32_VECTOR_PEAK_TFLOPS, "frac": tflops / FP32_VECTOR_PEAK_TFLOPS}


def _oracle_worker(job):
    """(spawned process) run the oracle over a slab of blocks; returns (n, seconds)."""
    os.environ["OMP_NUM_THREADS"] = "1"
    n, h, blocks, template, window = job
    orc = make_oracle(n, h, template, 0, window)
    t0 = time.perf_counter()
    for i in range(len(blocks)):
        orc.detect_u8(i, blocks[i])
    return len(blocks), time.perf_counter() - t0


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), or None if unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def physical_cores():
    """(worker count, how it was derived): one per physical core of the CPUs this process may use,
    capped by the container's CPU quota (more workers than quota only time-slice)."""
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = set()
    for c in avail:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                cores.add(f.read().strip())
        except OSError:
            cores.add(str(c))
    n, how = max(1, len(cores)), "%d logical CPUs available, %d distinct SMT sibling sets" % (len(avail), len(cores))
    q = cpu_quota()
    if q is not None and q < n:
        n, how = max(1, int(q)), how + "; cgroup cpu.max allows %.1f CPUs -> %d workers" % (q, max(1, int(q)))
    return n, how


def cpu_all_cores(n, h, blocks_u8, template, procs, single_rate, seconds=4.0, window=WINDOW_BINS):
    """Oracle throughput with `procs` spawned workers (one per physical core), each over its own
    slab of the blocks, sized from the single-core rate so the leg takes `seconds` if the cores
    scaled perfectly (they do not: all-core clocks and memory bandwidth; expect ~2x that)."""
    import multiprocessing as mp
    # one compute thread per worker: the children read these when THEY import NumPy / SciPy
    # (set inside the worker it would be too late -- the BLAS thread pools start at import)
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    per_proc = int(max(4, min(len(blocks_u8), single_rate * seconds)))
    jobs = [(n, h, blocks_u8[(i * per_proc) % max(1, len(blocks_u8) - per_proc + 1):][:per_proc], template,
             tuple(window)) for i in range(procs)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        pool.map(_oracle_worker, [(n, h, j[2][:2], template, tuple(window)) for j in jobs])   # warm imports
        t0 = time.perf_counter()
        done = pool.map(_oracle_worker, jobs)
        dt = time.perf_counter() - t0
    nb = sum(d[0] for d in done)
    return {"value": nb / dt, "unit": "blocks/s", "procs": procs, "blocks": nb, "seconds": dt}


def card_to_toad_leg(n_card):
    """BASELINE configs[0]: example detector.cfg + example template on a synthetic .card stream,
    the whole plumbing path text -> parse -> detect -> .toad text.  CPU = the oracle (what the
    reference does per line: base64 decode, (u8 - 127.4)/128, Detector.detect, serialize), one
    core; GPU = thrifty_amd's `thrifty detect --quiet -o` path: Detector.write_toad(), which for a
    regular file is ONE library call (thr_run_card: host framing, on-device base64 decode, batched
    engine, .toad text formatted and written by a library thread).  The same for the raw u8 form
    of the stream (`--raw`, thr_run_stream).  The example template is a data fixture
    (tests/golden/c1.npz holds it together with the example settings)."""
    import base64
    import tempfile
    from oracle import thrifty_np as onp
    from thrifty_amd import block_data, synth
    from thrifty_amd.detect import Detector, DetectorSettings
    g = np.load(os.path.join(ROOT, "tests", "golden", "c1.npz"), allow_pickle=False)
    n, h, tpl = int(g["block_len"]), int(g["history_len"]), g["template"]
    cthr, cwin, xthr = tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]), tuple(g["corr_thresh"])
    rng = np.random.default_rng(SEED + 1)
    ook = (tpl - tpl.min()) / (tpl.max() - tpl.min()) * 2 - 1     # synth draws 0.3 * (t + 1) / 2
    seed_blocks, _ = synth.synth_blocks(rng, 64, n, ook, onp.unique_window(n, h, len(tpl)))
    payload = [block_data.card_line(0.0, 0, seed_blocks[j]).split(" ", 2)[2] for j in range(64)]

    def card_lines(lo, hi):
        return "".join("%.6f %d %s" % (1000.0 + 0.005 * i, i, payload[i % 64]) for i in range(lo, hi)).encode()

    # --- CPU: one core, the reference's per-line loop
    orc = onp.OracleDetector(n, h, tpl, cthr, cwin, xthr)
    n_cpu = min(n_card, 256)
    lines = card_lines(0, n_cpu).split(b"\n")[:n_cpu]
    t0 = time.perf_counter()
    cpu_out = []
    for ln in lines:
        ts, idx, enc = ln.decode("ascii").split(" ")
        raw = np.frombuffer(base64.b64decode(enc), dtype=np.uint8)
        (res,) = orc.detect_u8(int(idx), raw)
        if res.detected:
            cpu_out.append(onp.toad_line(0, float(ts), int(idx), res))
    t_cpu = time.perf_counter() - t0
    st = DetectorSettings(n, h, len(tpl), cthr, cwin, tpl, xthr)

    def run_file(path, reader, out_path):
        """`thrifty detect --quiet <path> -o <out_path>`: -> (seconds incl. Detector construction,
        the library loop's own statistics)"""
        with open(path, "rb") as f, open(out_path, "wb") as out:
            t0 = time.perf_counter()      # (opening the reader and the engine handle is part of the job)
            det = Detector(st, reader(f), rxid=0)
            t1 = time.perf_counter()
            stats = det.write_toad(out)
            out.flush()
            dt = time.perf_counter() - t0
            det.close()     # (now, not at garbage collection: the next Detector reuses the runtime's pools)
            if stats is not None:
                stats["construct_s"], stats["write_toad_s"] = t1 - t0, dt - (t1 - t0)
            return dt, stats

    def iterate_file(path, reader, serialize):
        """The reference's operator loop itself (detect.py:217-223): `for detected, result in
        Detector(settings, card_reader(f))` -- one (detected, DetectionResult) per input block; with
        `serialize` every detection's .toad line too (EVERY block of this dense file: the worst case
        -- a receiver's capture holds a few detections per second).  -> (seconds incl.
        construction, blocks, detections, .toad lines)"""
        with open(path, "rb") as f:
            t0 = time.perf_counter()
            det = Detector(st, reader(f), rxid=0)
            n_blocks, hits, lines = 0, 0, []
            for detected, result in det:
                n_blocks += 1
                if detected:
                    hits += 1
                    if serialize:
                        lines.append(result.serialize())
            dt = time.perf_counter() - t0
            det.close()
            return dt, n_blocks, hits, lines

    out = {"config": "BASELINE configs[0]: example detector.cfg settings (block 16384, history %d, %d-sample "
                     "template, window bins %d..%d) on a synthetic .card stream" % (h, len(tpl), cwin[0], cwin[1]),
           "cpu_blocks_per_s": n_cpu / t_cpu, "cpu_blocks": n_cpu, "cpu_cores": 1}
    with tempfile.TemporaryDirectory() as tmpd:
        card, toad = (os.path.join(tmpd, x) for x in ("rx.card", "rx.toad"))
        def settle(f):
            """The file as a capture written long ago is: its pages clean in the page cache (locking
            pages that are still under write-back is slower, and says nothing about the detector)."""
            f.flush()
            os.fsync(f.fileno())

        text_bytes = 0
        with open(card, "wb") as f:      # a regular file, as `thrifty detect rx.card`
            for lo in range(0, n_card, 4096):
                text_bytes += f.write(card_lines(lo, min(n_card, lo + 4096)))
            settle(f)
        # warm-up, not timed: one pass of the same job, as W warm-up steps precede the timed steps of
        # the main leg.  (A whole pass, not a few batches: the HIP runtime keeps growing its pools for
        # the first ~30 batches of a process -- 8-15 ms stalls in the first copies of records out of
        # each pipeline slot; a pass over a DIFFERENT, never-read file in a warm process runs at the
        # rate of the second pass over this one, profiles/README.md round 5.)
        run_file(card, lambda f: block_data.CardStream(f, n), os.path.join(tmpd, "warm.toad"))
        t_gpu, stats = run_file(card, lambda f: block_data.CardStream(f, n), toad)
        gpu_out = open(toad, "rb").read().decode("ascii").split("\n")[:-1]   # (for the check below, not timed)
        loop = (stats or {}).get("calls", [{}])[-1]
        out.update({"gpu_blocks_per_s": n_card / t_gpu, "gpu_blocks": n_card,
                    # the library loop alone (thr_run_card's own clock: no Detector construction)
                    "gpu_loop_blocks_per_s": (loop["blocks"] / loop["total_s"]) if loop.get("total_s") else None,
                    "gpu_loop_stats": {k: loop.get(k) for k in ("batches", "total_s", "frame_s", "submit_s", "wait_s",
                                                                "format_s", "write_s", "window", "submit_phases")},
                    "gpu_construct_s": (stats or {}).get("construct_s"),
                    "gpu_includes": "a %.1f GB file in the page cache, after one untimed pass of the same job; "
                                    "Detector construction, thr_run_card (host framing, H2D of the base64 "
                                    "text out of the page-locked input window, device decode, detection, D2H, "
                                    ".toad text formatted and written by a library thread), file on disk"
                                    % (text_bytes / 1e9),
                    "detections_cpu": len(cpu_out), "detections_gpu": len(gpu_out)})
        # --- the object-building iteration over the same file: every block a (detected, DetectionResult)
        # (the faster of two passes after a warm-up pass, both times in the detail: this loop runs one
        # Python thread beside the input window's page-locking threads, and a pass that starts while the
        # previous detector's last pages are still being unlocked runs up to 25 % slower)
        iterate_file(card, lambda f: block_data.CardStream(f, n), False)
        passes = [iterate_file(card, lambda f: block_data.CardStream(f, n), False) for _ in range(2)]
        t_it, n_it, hits_it, _ = min(passes, key=lambda r: r[0])
        t_its, _, _, it_lines = iterate_file(card, lambda f: block_data.CardStream(f, n), True)
        out.update({"iter_blocks_per_s": n_it / t_it, "iter_blocks": n_it, "iter_detections": hits_it,
                    "iter_pass_seconds": [r[0] for r in passes],
                    "iter_serialize_blocks_per_s": n_it / t_its,
                    "iter_text_equals_write_toad": it_lines == gpu_out,
                    "iter_includes": "Detector construction, `for detected, result in Detector(settings, "
                                     "CardStream(f))` over the same file: one (detected, DetectionResult) per "
                                     "block (built a batch at a time by thrifty_amd._fastresults); "
                                     "iter_serialize: plus result.serialize() for every detection, here every block"})
        same = [a.split()[:3] + [a.split()[4], a.split()[8]] for a in gpu_out[:len(cpu_out)]] == \
               [b.split()[:3] + [b.split()[4], b.split()[8]] for b in cpu_out]
        out["outputs_agree"] = bool(same)       # rxid, time, block, sample, carrier bin of the first lines
        # --- the raw form of the same stream (`thrifty detect --raw`): the blocks' NEW samples back to
        # back, overlap framing on the device (thr_run_stream); the reference's zero-history lead-in
        # goes through the complex64 path first
        step = 2 * (n - h)
        rawp = os.path.join(tmpd, "rx.bin")
        n_raw = n_card
        with open(rawp, "wb") as f:
            chunk = np.concatenate([seed_blocks[j][-step:] for j in range(64)]).tobytes()
            for _ in range(n_raw // 64):
                f.write(chunk)
            settle(f)
        run_file(rawp, lambda f: block_data.RawStream(f, n, h), os.path.join(tmpd, "warm2.toad"))
        t_raw, rstats = run_file(rawp, lambda f: block_data.RawStream(f, n, h), os.path.join(tmpd, "raw.toad"))
        rloop = (rstats or {}).get("calls", [{}])[-1]
        iterate_file(rawp, lambda f: block_data.RawStream(f, n, h), False)
        rpasses = [iterate_file(rawp, lambda f: block_data.RawStream(f, n, h), False) for _ in range(2)]
        t_rit, n_rit, _, _ = min(rpasses, key=lambda r: r[0])
        t_rits, _, _, rit_lines = iterate_file(rawp, lambda f: block_data.RawStream(f, n, h), True)
        raw_text = open(os.path.join(tmpd, "raw.toad"), "rb").read().decode("ascii").split("\n")[:-1]
        out.update({"raw_iter_blocks_per_s": n_rit / t_rit, "raw_iter_blocks": n_rit,
                    "raw_iter_pass_seconds": [r[0] for r in rpasses],
                    "raw_iter_serialize_blocks_per_s": n_rit / t_rits,
                    "raw_iter_text_equals_write_toad": rit_lines == raw_text})
        out.update({"raw_gpu_loop_stats": {k: rloop.get(k) for k in ("batches", "total_s", "frame_s", "submit_s", "wait_s",
                                                                     "format_s", "write_s", "window", "submit_phases")},
                    "raw_gpu_blocks_per_s": (rstats or {}).get("blocks", 0) / t_raw,
                    "raw_gpu_blocks": (rstats or {}).get("blocks", 0),
                    "raw_detections_gpu": (rstats or {}).get("detections", 0)})
    return out


def parallel_cpu_budget():
    from thrifty_amd import parallel
    return parallel.cpu_budget()


def preflight(torch, dist, dev, cdev, rank, world, local, total, first, shared_gpu=False, env_applied=None):
    """Fail early and legibly if the process group is not what the launch line says: RCCL sees
    `world` ranks, every rank has its own device, and the block ranges tile [0, world * total).
    (shared_gpu: the gloo rehearsal, where every rank computes on cuda:0 by design.)"""
    one = torch.ones(1, dtype=torch.int64, device=cdev)
    dist.all_reduce(one)
    if int(one.item()) != world:
        raise SystemExit("pre-flight: all_reduce over %s counted %d ranks, expected %d"
                         % (dist.get_backend(), int(one.item()), world))
    mine = {"rank": rank, "local_rank": local, "cuda": dev.index, "device": torch.cuda.get_device_name(dev),
            "pci": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None),
            "blocks": [first, first + total], "pid": os.getpid(), "env": env_applied or {},
            "cpus": parallel_cpu_budget()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        spans = sorted(e["blocks"] for e in everyone)
        ok = spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        if not ok or (not shared_gpu and len({e["local_rank"] for e in everyone}) != world):
            raise SystemExit("pre-flight: ranks do not tile the block range / share a device: %r" % (everyone,))
        print("pre-flight ok: backend %s, %d rank(s); per-rank blocks: %s" % (
            dist.get_backend(), world, ", ".join("r%d@cuda:%d [%d, %d)" % (
                e["rank"], e["cuda"], e["blocks"][0], e["blocks"][1]) for e in everyone)),
            file=sys.stderr)
        # the environment every rank runs with (parallel.rank_env: the same on both launch routes)
        for e in everyone:
            print("pre-flight env: rank %d pid %d %s; %d CPUs for %d rank(s)" % (
                e["rank"], e["pid"], " ".join("%s=%s" % kv for kv in sorted(e["env"].items())),
                e["cpus"], world), file=sys.stderr)
        bad = [e["rank"] for e in everyone if e["env"] != everyone[0]["env"]]
        if bad:
            raise SystemExit("pre-flight: ranks %r run with a different environment than rank 0" % (bad,))


class Leg(object):
    """One workload resident in HBM on one GPU: templates, engine handle(s), synthetic u8 blocks,
    record buffers -- and the loops that run launch batches over them."""

    def __init__(self, torch, F, synth, dev, local, name, T=1, mix="dense", batch=0, resident=0, streams=0,
                 pnum=0, seed_off=0, first=0, carrier_window=WINDOW_BINS):
        cfg = CONFIGS[name]
        self.torch, self.F, self.dev, self.name, self.cfg = torch, F, dev, name, cfg
        self.n, self.h, self.T, self.mix, self.pnum = cfg["n"], cfg["h"], T, mix, pnum
        self.B = batch or cfg["batch"]
        self.thresh = (THRESH, THRESH)
        if "fixture" in cfg:
            # the reference's example template + settings (a data fixture); the synthetic bursts use
            # the template rescaled to [-1, 1] the way SURVEY 8(d)'s generator uses a +-1 code
            g = np.load(os.path.join(ROOT, "tests", "golden", cfg["fixture"]), allow_pickle=False)
            if T != 1:
                raise SystemExit("config %s holds one template" % name)
            tpl = np.asarray(g["template"], dtype=np.float64)
            self.tpls = tpl[None, :]
            synth_tpl = (tpl - tpl.min()) / (tpl.max() - tpl.min()) * 2 - 1
            self.thresh = (tuple(float(v) for v in g["carrier_thresh"]), tuple(float(v) for v in g["corr_thresh"]))
        else:
            self.tpls = np.stack([synth.gold_template(cfg["bits"], 2 + i, cfg["sps"]) for i in range(T)]).astype(np.float64)
            synth_tpl = self.tpls[0]
        self.wlen = self.tpls.shape[1]
        pad = self.h - self.wlen + 1
        self.window = (pad // 2, (self.n - self.wlen + 1) - (pad - pad // 2))
        n_handles = max(1, streams or cfg["streams"])
        self.cwin = tuple(carrier_window)
        self.sectioned = self.n > 16384 and not pnum and bool(F.plan_sections(self.n, self.h, self.wlen))
        self.engs = [F.Engine(self.n, self.h, self.tpls, self.thresh[0], self.cwin, self.thresh[1], device_id=local,
                              max_batch=self.B, preshift_num=pnum) for _ in range(n_handles)]
        # block_len 16384, short template(s): the correlate slot times k_correlate_4k over the (block,
        # 4096-sample section) items (csrc/detect16k_sec.hip); the engine says whether it does
        self.sec4k = self.engs[0].sections()[1] == 4096
        self.n_sec = self.engs[0].sections()[0]
        self.resident_batches = max(1, (resident or cfg["resident"]) // self.B)
        self.total = self.resident_batches * self.B
        self.first = first
        gen = torch.Generator(device=dev)
        gen.manual_seed(SEED + cfg["idx"] + 1 + seed_off)
        t0 = time.perf_counter()
        self.data = synth_on_device(torch, dev, gen, self.total, self.n, synth_tpl, self.window,
                                    1.0 if mix == "dense" else 0.1)
        torch.cuda.synchronize()
        self.t_gen = time.perf_counter() - t0
        self.idx = torch.arange(first, first + self.total, dtype=torch.int64, device=dev)
        self.rec = torch.zeros((self.total * T, 64), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()     # fills done before the engines' own streams write records
        self.bytes_per_block = 2 * self.n + 64 * T

    def batch(self, g, eng=None):
        """launch batch number g (cycles over the resident blocks, alternates over the handles)"""
        s = (g % self.resident_batches) * self.B
        (eng or self.engs[g % len(self.engs)]).detect_device(
            self.data[s:s + self.B].data_ptr(), self.F.THR_IN_U8, self.B,
            self.rec[s * self.T:].data_ptr(), self.idx[s:].data_ptr())

    def sync(self):
        for e in self.engs:
            e.sync()

    def run(self, g0, n_batches, solo_every=0, engs=None):
        """n_batches launch batches starting at number g0; with solo_every > 0 every
        solo_every-th batch runs ALONE with HIP events around its kernels (the roofline leg:
        kernel durations measured while two batches overlap would be inflated by the sharing).
        Not synchronised at the end.  -> number of solo (profiled) batches"""
        engs = engs or self.engs
        solo = 0
        for g in range(g0, g0 + n_batches):
            e = engs[g % len(engs)]
            if solo_every > 0 and g % solo_every == 0:
                for x in engs:
                    x.sync()
                e.profile_enable(1)
                self.batch(g, e)
                e.sync()
                e.profile_enable(0)
                solo += 1
            else:
                self.batch(g, e)
        return solo

    def timed(self, g0, n_batches, solo_every=0, engs=None):
        """run() bracketed by device synchronisation -> (seconds, solo batches)"""
        self.sync()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        solo = self.run(g0, n_batches, solo_every, engs)
        for x in (engs or self.engs):
            x.sync()
        return time.perf_counter() - t0, solo

    def reset_profile(self):
        for e in self.engs:
            e.profile_enable(0)
            e.profile_read()

    def read_profile(self):
        prof = {}
        for e in self.engs:
            for k, (ms, cnt) in e.profile_read().items():
                prof[k] = (prof.get(k, (0.0, 0))[0] + ms, prof.get(k, (0.0, 0))[1] + cnt)
            e.profile_enable(0)
        rename = {}
        if self.pnum:   # the fused kernel is timed in k_correlate's event slot
            rename["k_correlate"] = "k_preshift"
        if self.sec4k:
            rename["k_correlate"] = "k_correlate_4k"
        if self.n > 16384:
            # long blocks: the correlate slot times k_correlate<SEG> over the (block, section) items --
            # or, for templates too long to section, the fused sub-transform + combination kernel
            rename.update({"k_correlate": "k_correlate_seg" if self.sectioned else "k_correlate_sub",
                           "k_carrier": "k_carrier_dit+k_select_dit"})
        return {rename.get(k, k): v for k, v in prof.items() if v[1] > 0}

    def roofline(self, prof, solo_batches, fallback_ms, rate_per_gpu=None):
        """-> (roofline, detail).  `roofline`: the recomputable object of the dominant kernel --
        algorithmic bytes per launch (2N + 64 T per block x blocks per launch) / its mean
        launch duration (HIP events on the launch stream) -- flat scalars only, so that the driver's
        record keeps all of it.  `detail`: every kernel's mean duration, launches per batch and the
        VALU view; it goes to the detail file / stderr, not into the contract line."""
        dom = max(prof, key=lambda k: prof[k][0]) if prof else None
        dom_ms, dom_cnt = prof[dom] if prof else (0.0, 0)
        if dom_cnt == 0:   # no profiled batch: fall back to the whole launch batch
            dom, avg_ms, units = "all kernels of one launch batch", fallback_ms, float(self.B)
        else:
            avg_ms = dom_ms / dom_cnt
            units = self.B * max(solo_batches, 1) / dom_cnt      # blocks one launch of it processes
        achieved = self.bytes_per_block * units / (avg_ms * 1e-3) / 1e9
        # HBM traffic from the committed PMC passes of this workload (profiles/hbm_traffic.json,
        # scripts/profile_gpu.sh): the dominant kernel's, and the sum over every kernel of a launch
        # batch (`pipeline_traffic`: each stage fetches the block again).  The file names the hash
        # of the kernel sources it was measured on; `traffic_stale` says it is not today's.
        traffic = clock = pipeline = stale = src = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                per_kernel = json.load(open(tpath)).get(self.traffic_key(), {})
                # (the engine's carrier slot times whichever carrier kernel the window selects: the
                # pruned one for the 7..110 window, the full-spectrum one for (0, -1))
                alias = {"k_carrier": ["k_carrier" if self.cwin != tuple(WINDOW_BINS) else "k_carrier_pruned"],
                         "k_carrier_dit+k_select_dit": ["k_carrier_dit", "k_select_dit"]}
                def entries(k):
                    return [per_kernel[a] for a in alias.get(k, [k]) if a in per_kernel]
                if per_kernel:
                    e = entries(dom)
                    traffic = sum(x["bytes_per_launch"] for x in e) if e else None
                    clock = e[0].get("effective_clock_ghz") if e else None
                    every = [x for k in prof for x in entries(k)]
                    if every and all(entries(k) for k in prof):
                        pipeline = sum(x["bytes_per_launch"] for x in every)
                    src = per_kernel.get("_source", {})
                    stale = src.get("csrc_sha16") != csrc_sha16()
            except Exception:
                traffic = clock = pipeline = stale = None
        alg_launch = self.bytes_per_block * units
        comp = _compute_view(dom, self.n, self.T, units, avg_ms, self.n_sec)
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": avg_ms, "launches": dom_cnt, "blocks_per_launch": units,
                "algorithmic_bytes_per_block": self.bytes_per_block,
                "traffic_over_algorithmic": (traffic / alg_launch) if traffic else None,
                # all kernels of one launch batch (carrier stage + fit + correlate + finish) over the
                # batch's algorithmic bytes: 1.0 if every block were fetched exactly once
                "pipeline_traffic_over_algorithmic": (pipeline / (self.bytes_per_block * float(self.B))
                                                      if pipeline else None),
                # the WHOLE pipeline against the HBM peak: this leg's blocks/s (per GPU, wall clock)
                # x algorithmic bytes per block / 8 TB/s -- what the path as a whole achieves, beside
                # the dominant kernel's own `frac`
                "pipeline_frac": (rate_per_gpu * self.bytes_per_block / 1e9 / HBM_PEAK_GBS
                                  if rate_per_gpu else None),
                # the kernels are VALU/LDS-bound (about 80 flop per input byte): nominal 5 N log2 N flop
                # per transform this kernel performs (+ 6N per pointwise product) / fp32 vector peak
                "valu_frac": comp["frac"],
                "traffic_source": (src or {}).get("summary"), "traffic_stale": stale}
        detail = {"kernel": dom, "pipeline_traffic": pipeline, "traffic": traffic,
                  # GRBM_GUI_ACTIVE / duration of this kernel in the committed PMC pass (profiles/): the
                  # clock the part sustained under it (MI355X_MICROARCH.md, "DVFS give-back"); max 2.4
                  "effective_clock_ghz_profiled": clock,
                  "algorithmic_bytes_per_launch": alg_launch,
                  "all_kernels_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items()},
                  "all_kernels_launches_per_batch": {k: v[1] / max(solo_batches, 1) for k, v in prof.items()},
                  "compute": comp}
        return roof, detail

    def traffic_key(self):
        """This workload's key in profiles/hbm_traffic.json (scripts/collect_profiles.py)."""
        return (self.name + ("_t%d" % self.T if self.T > 1 else "") + ("_sparse" if self.mix != "dense" else "")
                + ("_fullwin" if self.cwin != tuple(WINDOW_BINS) else ""))

    def workload_text(self):
        return ("BASELINE configs[%d]: %s, %s, %d blocks/GPU resident in HBM as u8 IQ%s"
                % (4 if (self.T > 1 and self.n == 16384) else self.cfg["idx"], self.cfg["label"], self.mix,
                   self.total, "" if self.cwin == tuple(WINDOW_BINS) else
                   "; carrier window %d..%d (the reference's default '0--1': every bin)" % self.cwin))

    def host_records(self, nblocks):
        return self.rec.view(self.total, self.T, 64)[:nblocks, 0].cpu().numpy().view(self.F.RECORD_DTYPE).reshape(-1)

    def close(self):
        for e in self.engs:
            e.close()
        self.engs = []
        del self.data, self.rec, self.idx


def _parity_worker(job):
    """(spawned process) oracle over a slab of blocks -> rows of the exact fields + floats"""
    os.environ["OMP_NUM_THREADS"] = "1"
    n, h, blocks, template, lo = job
    orc = make_oracle(n, h, template)
    out = []
    for i in range(len(blocks)):
        (res,) = orc.detect_u8(lo + i, blocks[i])
        c = res.corr
        out.append((res.carrier.bin, bool(res.carrier.detected), int(c.sample) if c else -1,
                    bool(c.detected) if c else False, float(c.energy) if c else 0.0,
                    float(c.offset) if c else 0.0))
    return lo, out


def oracle_parity(n, h, blocks_u8, template, gpu_rec, procs):
    """GPU records against the oracle over `blocks_u8`, the oracle spread over `procs` spawned
    workers (checker only: nothing here is timed as the product)."""
    import multiprocessing as mp
    from thrifty_amd import _native as F
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    chunk = max(1, min(64, len(blocks_u8) // max(1, 2 * procs)))
    jobs = [(n, h, blocks_u8[s:s + chunk], template, s) for s in range(0, len(blocks_u8), chunk)]
    t0 = time.perf_counter()
    rows = [None] * len(blocks_u8)
    with mp.get_context("spawn").Pool(procs) as pool:
        for lo, out in pool.imap_unordered(_parity_worker, jobs):
            rows[lo:lo + len(out)] = out
    dt = time.perf_counter() - t0
    mism = 0
    for i, (cbin, cdet, samp, det, en, off) in enumerate(rows):
        r = gpu_rec[i]
        ok = r["carrier_bin"] == cbin and bool(r["flags"] & F.FLAG_CARRIER) == cdet
        if ok and cdet:
            ok = (r["corr_sample"] == samp and bool(r["flags"] & F.FLAG_CORR) == det and
                  abs(r["corr_energy"] - en) <= 1e-4 * abs(en) and abs(r["corr_offset"] - off) <= 1e-4 + 1e-4 * abs(off))
        mism += 0 if ok else 1
    return {"value": len(rows) / dt, "unit": "blocks/s", "cores": procs, "kind": "port",
            "sample": "%d blocks of one launch batch of this leg through oracle/thrifty_np.py, %d spawned "
                      "workers (pool start-up included), %.1f s" % (len(rows), procs, dt),
            "parity_checked": len(rows), "parity_mismatches": mism,
            "parity_fields": "carrier bin, both verdicts, SoA sample index exact; corr energy and sub-sample "
                             "offset 1e-4"}


def extra_leg(torch, F, synth, dev, local, key, seconds, cpu_ok):
    """One bounded leg of the default run: another single-GPU workload of BASELINE.json's configs,
    timed over ~`seconds` after a calibration burst.  -> (compact object for the line, detail)."""
    spec = LEGS[key]
    t_leg = time.perf_counter()
    leg = Leg(torch, F, synth, dev, local, spec["name"], T=spec["T"], mix=spec["mix"], batch=spec["batch"],
              resident=spec["resident"], seed_off=17, carrier_window=spec.get("window", WINDOW_BINS))
    leg.timed(0, 2)                                           # code load, first-touch
    dt, _ = leg.timed(0, 4)                                   # calibration burst
    nb = int(max(8, min(4096, seconds / (dt / 4))))
    nb -= nb % len(leg.engs)
    leg.reset_profile()
    solo_every = max(2, nb // 6)
    if len(leg.engs) == 1:        # one handle: nothing overlaps, HIP events around every solo_every-th batch
        leg.engs[0].profile_enable(solo_every)
        dt, _ = leg.timed(0, nb, 0)
        solo = (nb + solo_every - 1) // solo_every
    else:
        dt, solo = leg.timed(0, nb, solo_every)
    prof = leg.read_profile()
    value = nb * leg.B / dt
    roof, rdetail = leg.roofline(prof, solo, dt / nb * 1e3, value)
    keep = ("kernel", "frac", "avg_launch_ms", "blocks_per_launch", "algorithmic_bytes_per_block",
            "traffic_over_algorithmic", "pipeline_traffic_over_algorithmic", "pipeline_frac", "valu_frac",
            "traffic_stale")
    out = {"value": value, "blocks_per_launch_batch": leg.B, "templates": leg.T,
           "handles_per_gpu": len(leg.engs)}
    out.update({k: roof[k] for k in keep})
    detail = {"workload": leg.workload_text(), "value": value, "unit": "blocks/s", "launch_batches": nb,
              "blocks": nb * leg.B, "seconds": dt, "ms_per_launch_batch": dt / nb * 1e3,
              "solo_profiled_batches": solo, "roofline": roof, "roofline_detail": rdetail,
              "data_gen_s": leg.t_gen}
    if key == "c3" and cpu_ok:
        # the oracle check at benchmark shape: 2048 blocks of ONE 16384-block launch batch
        ns = 2048
        procs, _ = physical_cores()
        par = oracle_parity(leg.n, leg.h, leg.data[:ns].cpu().numpy(), leg.tpls[0], leg.host_records(ns), procs)
        detail["cpu_baseline"] = par
        out["parity_checked"], out["parity_mismatches"] = par["parity_checked"], par["parity_mismatches"]
    leg.close()
    detail["leg_wall_s"] = time.perf_counter() - t_leg
    return out, detail


def write_detail(detail, tag):
    """Everything the contract line no longer carries (per-kernel times of every leg, the VALU view,
    workload prose, CPU-leg samples): gpurun_out/bench_detail_<tag>.json -- gpurun_out/ is what comes
    home from a GPU box -- and one line on stderr."""
    text = json.dumps(detail)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail_%s.json" % tag), "w") as f:
            f.write(text + "\n")
    except OSError:
        pass
    print("bench detail: " + text, file=sys.stderr)


def main():
    args = parse_args()
    # the rank environment FIRST, before torch / the HIP runtime initialise: the same on every launch
    # route (the driver's torch.distributed.run line, our own relaunch, a user's torchrun)
    from thrifty_amd import parallel
    env_applied = parallel.rank_env()
    import torch
    import torch.distributed as dist

    from thrifty_amd import _native as F
    from thrifty_amd import synth

    cfg = CONFIGS[args.config]
    n, h = cfg["n"], cfg["h"]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver would
        # (torch.distributed.run, one node, 127.0.0.1); rank 0 of the children prints the line
        raise SystemExit(parallel.relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    gloo = args.dist_backend == "gloo"
    if gloo:
        local = 0           # rehearsal: every rank computes on cuda:0
    elif local >= torch.cuda.device_count():
        raise SystemExit("rank %d wants cuda:%d but %d device(s) are visible (one GPU per rank; "
                         "--dist-backend gloo rehearses the N-rank body on one GPU)"
                         % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if gloo else dev      # where the collectives' tensors live
    # under torchrun (RANK set) always bring RCCL up, even for one rank: the record gather
    # then runs through the same collectives as the multi-GPU case
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        # (a rank that dies must not leave its peers waiting for the default half hour: five minutes
        # cover the slowest phase, 32 GiB of synthetic blocks per rank, many times over)
        import datetime
        limit = datetime.timedelta(minutes=5)
        if gloo:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)

    K, W, T = args.steps, args.warmup, args.templates
    pnum = args.preshift_num if args.variant == "preshift" else 0
    B = args.batch or cfg["batch"]
    strong = args.scaling == "strong"
    # distinct blocks resident in HBM: up to the config's limit (c2: 1 Mi blocks = 32 GiB of u8 =
    # BASELINE's "1M synthetic blocks"); longer runs cycle through them.  --scaling strong: the
    # config's blocks IN TOTAL (configs[3] as written: 1 Mi blocks sharded 8 ways), 1/world per rank
    resident = min(cfg["resident"], max(B, K * B)) if args.min_seconds <= 0 else cfg["resident"]
    if args.resident_blocks > 0:
        resident = max(B, args.resident_blocks)
    if strong:
        resident = max(B, resident // world)
    leg = Leg(torch, F, synth, dev, local, args.config, T=T, mix=args.mix, batch=B, resident=resident,
              streams=args.streams, pnum=pnum, seed_off=rank, first=0, carrier_window=tuple(args.carrier_window))
    total = leg.total
    # contiguous block-index range per rank
    first = rank * total
    leg.idx += first
    leg.first = first
    engs, eng = leg.engs, leg.engs[0]
    if len(engs) == 1:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    gather_method = None
    if use_dist:
        preflight(torch, dist, dev, cdev, rank, world, int(os.environ.get("LOCAL_RANK", "0")), total, first,
                  shared_gpu=gloo, env_applied=env_applied)
        # one tiny record gather on the real backend (uneven counts, one empty rank), outside the
        # timed region: a broken gather / padding path ends the run HERE with a sentence -- or the
        # all_gather form takes over for the run
        gather_method = parallel.gather_selftest(world, rank, cdev)
        if rank == 0:
            print("pre-flight gather selftest ok: method %s, counts %s" % (
                gather_method, parallel.selftest_counts(world)), file=sys.stderr)
    kept = torch.zeros_like(leg.rec)
    torch.cuda.synchronize()

    # ---- calibration burst = the round-1/2 protocol: 1 Mi blocks (32 launch batches) straight after
    # a two-batch code-load warm-up -- reported as `value_first_1Mi`, and it sizes a step
    leg.timed(0, 2)
    # (--min-seconds 0 -- profiler passes, quick A/B runs: no burst, a step is one launch batch)
    burst_batches = max(2, min(leg.resident_batches, (1 << 20) // B)) if args.min_seconds > 0 else 2
    burst_dt, _ = leg.timed(0, burst_batches)
    burst_rate = burst_batches * B / burst_dt
    R = 1
    if args.min_seconds > 0:
        R = int(max(1, np.ceil(args.min_seconds * burst_rate / (K * B))))
    if use_dist:    # every rank must run the same step size: take rank 0's
        r_t = torch.tensor([R], dtype=torch.int64, device=cdev)
        dist.broadcast(r_t, 0)
        R = int(r_t.item())
    R_one_gpu = R
    if strong:
        # the job is what ONE GPU would run for --min-seconds (K x R launch batches); N ranks split it
        R = max(1, -(-R // world))
    for i in range(W):
        leg.run(i * R, R)
    leg.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # ---- the timed region: EXACTLY K steps, each R launch batches of B blocks.  Steps alternate
    # over the engine handles (double buffering: the latency-bound k_fit and the launch gaps of
    # one batch hide under the other batch's kernels); with more than one handle every
    # `profile_kernels`-th launch batch runs SOLO with HIP events around its kernels (costs that
    # batch's overlap; inside the timed region).
    solo_every = args.profile_kernels if len(engs) > 1 else 0
    leg.reset_profile()
    if len(engs) == 1 and args.profile_kernels > 0:
        eng.profile_enable(args.profile_kernels)
    profiled = 0
    t0 = time.perf_counter()
    for i in range(K):
        profiled += leg.run(i * R, R, solo_every)
    leg.sync()
    if len(engs) == 1 and args.profile_kernels > 0:
        profiled = (K * R + args.profile_kernels - 1) // args.profile_kernels
    t_compute = time.perf_counter()
    # K7 + C1: compact detected records, gather them to rank 0 (the only collective)
    n_kept = eng.compact_device(leg.rec.data_ptr(), total * T, kept.data_ptr())     # (synchronises)
    t_compact = time.perf_counter()
    gathered = (parallel.gather_records(kept[:n_kept].to(cdev), world, rank, cdev, force=use_dist)
                if use_dist else kept[:n_kept])
    torch.cuda.synchronize()
    t_gather = time.perf_counter()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = leg.read_profile()
    # (after the clock stopped, reporting only) every rank's own times: a sub-linear curve can be
    # attributed to a slow rank's kernels, the compaction or the gather
    mine = {"n_kept": int(n_kept), "compute_s": t_compute - t0, "compact_ms": (t_compact - t_compute) * 1e3,
            "gather_ms": (t_gather - t_compact) * 1e3, "total_s": dt}
    ranks = [mine]
    if use_dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    per_rank = [r["n_kept"] for r in ranks]
    # ---- one handle alone (what the roofline's solo kernel times belong to), a short leg
    one_dt, one_n = None, 0
    if len(engs) > 1 and args.min_seconds > 0:
        one_n = int(max(8, min(K * R, 0.25 * args.min_seconds * burst_rate / B)))
        one_dt, _ = leg.timed(0, one_n, 0, engs[:1])

    if rank == 0:
        blocks_total = world * K * R * B
        value = blocks_total / dt
        metric = ("IQ blocks/sec (16384-sample, 1024-chip template)" if n == 16384 and args.config == "c2"
                  else "IQ blocks/sec (%d-sample, %d-sample template)" % (n, leg.wlen))
        roof, rdetail = leg.roofline(prof, profiled, dt / (K * R) * 1e3, value / world)
        line = {
            "metric": metric,
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": leg.workload_text(),
                       "name": args.config,
                       "variant": args.variant if not pnum else "preshift(num=%d)" % pnum,
                       "blocks_per_step_per_gpu": R * B, "launch_batches_per_step": R,
                       # (the calibrated step of ONE GPU's job; --scaling strong splits it over the ranks)
                       "launch_batches_per_step_one_gpu": R_one_gpu,
                       "blocks_per_launch_batch": B, "templates": T,
                       "carrier_window": list(leg.cwin), "thresholds": "15*snr",
                       "parallelism": "block-shard x%d" % world,
                       "dist_backend": ((("gloo (rehearsal: every rank on cuda:0)" if gloo else "nccl (RCCL)"))
                                        if use_dist else None),
                       "gather_method": gather_method,
                       "rank_env": env_applied,
                       "handles_per_gpu": len(engs),
                       "detections_gathered": int(gathered.shape[0]),
                       "detections_per_rank": per_rank,
                       # per rank: seconds until its kernels were done, compaction, gather (ms)
                       "per_rank_seconds": [round(r["compute_s"], 6) for r in ranks],
                       "per_rank_compact_ms": [round(r["compact_ms"], 3) for r in ranks],
                       "per_rank_gather_ms": [round(r["gather_ms"], 3) for r in ranks]},
            # rank 0's own compaction and gather inside the timed region
            "compact_ms": ranks[0]["compact_ms"], "gather_ms": ranks[0]["gather_ms"],
            # the sustained-rate protocol (wall clock over >= 1 M blocks after warm-up)
            "timed_region_s": dt, "blocks_timed": blocks_total,
            "value_first_1Mi": burst_rate if args.min_seconds > 0 else None,
            "value_1stream": (one_n * B / one_dt) if one_dt else None,
            "roofline": roof,
        }
        detail = {"main": {"workload": leg.workload_text(), "roofline_detail": rdetail, "data_gen_s": leg.t_gen,
                           "first_1Mi": {"blocks": burst_batches * B, "seconds": burst_dt,
                                         "note": "rank 0's first %d launch batches after a two-batch code-load "
                                                 "warm-up (the round-1/2 protocol), no gather" % burst_batches},
                           "one_stream": ({"blocks": one_n * B, "seconds": one_dt,
                                           "note": "one engine handle alone on rank 0, straight after the "
                                                   "timed region"} if one_dt else None),
                           "ranks": ranks},
                  "configs": {}}
        legs = []
        if args.legs == "auto":
            legs = (list(LEGS) if (world == 1 and args.config == "c2" and T == 1 and not pnum
                                   and args.mix == "dense") else [])
        elif args.legs != "none":
            legs = [x for x in args.legs.split(",") if x]
        host_blocks = gpu_rec = None
        if world == 1 and args.cpu_seconds > 0:
            ns = min(total, 16384 if n == 16384 else 2048)
            host_blocks = leg.data[:ns].cpu().numpy()
            gpu_rec = leg.host_records(ns)
        summary = {args.config + ("_t%d" % T if T > 1 else "") + ("_sparse" if args.mix != "dense" else ""): value}
        if legs:
            leg.close()      # the main leg's 32 GiB go before the other configs' data arrive
            torch.cuda.empty_cache()
            line["configs"] = {}
            for key in legs:
                line["configs"][key], detail["configs"][key] = extra_leg(
                    torch, F, synth, dev, local, key, args.leg_seconds, cpu_ok=args.cpu_seconds > 0)
                summary[key] = line["configs"][key]["value"]
                torch.cuda.empty_cache()
        if host_blocks is not None:
            cb = cpu_baseline(n, h, host_blocks, np.arange(first, first + len(host_blocks)),
                              leg.tpls[0], args.cpu_seconds, gpu_rec, pnum, leg.cwin)
            line["cpu_baseline"] = cb
            procs, how = physical_cores()
            if args.cpu_procs >= 0:
                procs, how = args.cpu_procs, "--cpu-procs"
            if procs > 0 and not pnum:
                ac = cpu_all_cores(n, h, host_blocks, leg.tpls[0], procs, cb["value"], window=leg.cwin)
                ac["procs_from"] = how
                detail["cpu_all_cores"] = ac
                # (flat: the driver's record keeps the scalars of this object)
                cb["all_cores_value"], cb["all_cores_procs"] = ac["value"], ac["procs"]
            if args.card_blocks > 0 and args.config == "c2" and T == 1 and not pnum:
                if leg.engs:     # free the benchmark's engines before the plumbing legs create their own
                    leg.close()
                ct = card_to_toad_leg(args.card_blocks)
                detail["card_to_toad"] = ct
                for k in ("cpu_blocks_per_s", "gpu_blocks_per_s", "gpu_blocks", "gpu_loop_blocks_per_s",
                          "raw_gpu_blocks_per_s", "outputs_agree", "iter_blocks_per_s", "raw_iter_blocks_per_s",
                          "iter_serialize_blocks_per_s", "iter_text_equals_write_toad"):
                    if k in ct:
                        cb["card_to_toad_" + k] = ct[k]
                summary["card_to_toad"] = ct["gpu_blocks_per_s"]
                if "raw_gpu_blocks_per_s" in ct:
                    summary["raw_to_toad"] = ct["raw_gpu_blocks_per_s"]
                # the reference's own `for detected, result in Detector(...)` loop over the same files
                summary["detector_iter"] = ct["iter_blocks_per_s"]
                summary["detector_iter_raw"] = ct["raw_iter_blocks_per_s"]
        # flat, last: the one place where every leg's blocks/s stands side by side (the driver keeps
        # the tail of stdout)
        line["summary"] = {k: round(v) for k, v in summary.items()}
        write_detail(detail, "%s_n%d" % (args.config, world))
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
