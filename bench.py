#!/usr/bin/env python3
"""Benchmark of the `thrifty detect` hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mix dense|sparse]

One "step" = one pass of the hot path (FFT -> carrier detect -> fit -> shift ->
FFT -> x conj(T) -> IFFT -> SoA) over one batch of B synthetic IQ blocks that
are already resident in HBM.  Workload = BASELINE.json configs[1]:
block_len 16384, history 4096, 1023-chip Gold template, K*B (default
128 * 8192 = 1 Mi) blocks per GPU.  Metric: IQ blocks/s, whole job.

For N > 1 launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Blocks are sharded by contiguous block-index ranges (weak scaling: K*B blocks
per GPU); the only collective is the gather of detection records to rank 0.
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

N_BLOCK = 16384
HISTORY = 4096
SEED = 20260928 + 2
RESIDENT_BLOCKS = 2 << 20
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32768,
                    help="blocks per step (per GPU); the default x 32 steps = BASELINE's 1 Mi blocks")
    ap.add_argument("--mix", choices=["dense", "sparse"], default="dense",
                    help="dense: every block carries a signal; sparse: 10%% do")
    ap.add_argument("--templates", type=int, default=1)
    ap.add_argument("--streams", type=int, default=2,
                    help="engine handles (each with its own HIP stream) the steps alternate over")
    ap.add_argument("--profile-kernels", type=int, default=16,
                    help="n > 0: HIP events around the kernels of every n-th step of the timed "
                         "region (roofline leg; 1 = every step, costs ~4%%); 0 = off")
    ap.add_argument("--variant", choices=["default", "preshift"], default="default",
                    help="default: reference Detector (the headline); preshift: the reference's "
                         "experimental PreshiftDetector (one fused kernel per block)")
    ap.add_argument("--preshift-num", type=int, default=21, help="bank size of --variant preshift")
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="n > 0: also time the CPU oracle on n worker processes (spawned; adds "
                         "cpu_baseline.all_cores; off by default to keep the default run short)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="wall budget of the CPU baseline leg (0 disables)")
    return ap.parse_args()


def synth_on_device(torch, dev, gen, n_blocks, template, window, signal_frac, chunk=2048,
                    truth=None):
    """SURVEY.md 8(d) generator, on the GPU: OOK burst 0.3*(t+1)/2 at a uniform lag in
    the unique window, carrier bin ~ U(10,100), AWGN sigma 0.02, u8 quantiser
    (x*128 + 127.4, truncating).  Returns uint8 [n_blocks, 2N]; if `truth` is a dict it
    receives the drawn lag / carrier bin / has-signal tensors (tests/test_gpu_fullsize.py)."""
    n, w = N_BLOCK, len(template)
    lo, hi = window
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


def cpu_baseline(blocks_u8, idx, template, budget_s, gpu_rec, n_templates, preshift_num=0):
    """Oracle (oracle/thrifty_np.py, a NumPy port of the reference algorithm) timed on
    host cores over a bounded sample of the same blocks; also spot-checks parity."""
    from oracle import thrifty_np as onp
    from thrifty_amd import _native as F
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    if preshift_num:
        orc = onp.OraclePreshiftDetector(N_BLOCK, HISTORY, template, (0, 15, 0), (7, 110), (0, 15, 0),
                                         num=preshift_num)
    else:
        orc = onp.OracleDetector(N_BLOCK, HISTORY, template, (0, 15, 0), (7, 110), (0, 15, 0))
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
        if time.perf_counter() - t0 > budget_s and done >= 64:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "blocks/s", "cores": 1, "kind": "port",
            "sample": "%d of the benchmark's own u8 blocks (single-template) through "
                      "oracle/thrifty_np.py (NumPy %s pocketfft + SciPy curve_fit), 1 thread, %.1f s"
                      % (done, np.__version__, dt),
            "parity_checked": done, "parity_mismatches": mism}


FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X packed-fp32 VALU peak (MI355X_MICROARCH.md)


def _compute_view(kernel, n_templates, blocks_per_launch, avg_ms):
    """Nominal flop of the dominant kernel per block -> achieved TFLOP/s vs the VALU peak."""
    fft = 5.0 * N_BLOCK * np.log2(N_BLOCK)
    point = 6.0 * N_BLOCK
    if kernel == "k_correlate":      # shift, FFT#2, then per template: product, IFFT, |.|^2
        flop = point + fft + n_templates * (point + fft + 3.0 * N_BLOCK)
    elif kernel == "k_preshift":     # FFT#1, |X|^2, product, IFFT, |.|^2
        flop = fft + 3.0 * N_BLOCK + point + fft + 3.0 * N_BLOCK
    else:                            # carrier stage: FFT#1 (pruned variants do less) + |X|^2
        flop = fft + 3.0 * N_BLOCK
    tflops = flop * blocks_per_launch / (avg_ms * 1e-3) / 1e12
    return {"flop_per_block_nominal": flop, "achieved_tflops": tflops,
            "peak_tflops": FP32_VECTOR_PEAK_TFLOPS, "frac": tflops / FP32_VECTOR_PEAK_TFLOPS}


def _oracle_worker(job):
    """(spawned process) run the oracle over a slab of blocks; returns (n, seconds)."""
    os.environ["OMP_NUM_THREADS"] = "1"
    blocks, template = job
    from oracle import thrifty_np as onp
    orc = onp.OracleDetector(N_BLOCK, HISTORY, template, (0, 15, 0), (7, 110), (0, 15, 0))
    t0 = time.perf_counter()
    for i in range(len(blocks)):
        orc.detect_u8(i, blocks[i])
    return len(blocks), time.perf_counter() - t0


def cpu_all_cores(blocks_u8, template, procs, per_proc=384):
    """Oracle throughput with `procs` spawned workers, each over its own slab of the blocks."""
    import multiprocessing as mp
    jobs = [(blocks_u8[(i * per_proc) % len(blocks_u8):][:per_proc], template) for i in range(procs)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        pool.map(_oracle_worker, [(j[0][:4], template) for j in jobs])   # warm imports
        t0 = time.perf_counter()
        done = pool.map(_oracle_worker, jobs)
        dt = time.perf_counter() - t0
    n = sum(d[0] for d in done)
    return {"value": n / dt, "unit": "blocks/s", "procs": procs, "blocks": n, "seconds": dt}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from thrifty_amd import _native as F
    from thrifty_amd import parallel, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # under torchrun (RANK set) always bring RCCL up, even for one rank: the record gather
    # then runs through the same collectives as the multi-GPU case
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    B, K, W = args.batch, args.steps, args.warmup
    T = args.templates
    tpls = np.stack([synth.gold_template(10, 2 + i) for i in range(T)]).astype(np.float64)
    wlen = tpls.shape[1]
    pad = HISTORY - wlen + 1
    window = (pad // 2, (N_BLOCK - wlen + 1) - (pad - pad // 2))

    pnum = args.preshift_num if args.variant == "preshift" else 0
    engs = [F.Engine(N_BLOCK, HISTORY, tpls, (0, 15, 0), (7, 110), (0, 15, 0), device_id=local,
                     max_batch=B, preshift_num=pnum) for _ in range(max(1, args.streams))]
    eng = engs[0]
    if len(engs) == 1:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)

    # distinct blocks resident in HBM: every step has its own batch up to RESIDENT_BLOCKS
    # (2 Mi blocks = 64 GiB of u8); longer runs cycle through them
    resident_steps = max(1, min(K, RESIDENT_BLOCKS // B))
    total = resident_steps * B
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + rank)
    frac = 1.0 if args.mix == "dense" else 0.1
    t_gen = time.perf_counter()
    data = synth_on_device(torch, dev, gen, total, tpls[0], window, frac)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    # contiguous block-index range per rank
    first = rank * total
    idx = torch.arange(first, first + total, dtype=torch.int64, device=dev)
    rec = torch.zeros((total * T, 64), dtype=torch.uint8, device=dev)
    kept = torch.zeros_like(rec)

    def step(i):
        s = (i % resident_steps) * B
        engs[i % len(engs)].detect_device(data[s:s + B].data_ptr(), F.THR_IN_U8, B,
                                          rec[s * T:].data_ptr(), idx[s:].data_ptr())

    def sync_engines():
        for e in engs:
            e.sync()

    for i in range(W):
        step(i)
    sync_engines()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # Steps alternate over the engine handles (double buffering: the latency-bound k_fit and
    # the launch gaps of one batch hide under the other batch's kernels).  Kernel durations
    # measured while two batches overlap would be inflated by the sharing, so with more than
    # one handle the roofline leg times SOLO steps: every `profile_kernels`-th step of the
    # timed region runs with the other handle drained (costs the overlap of that step).
    solo = len(engs) > 1 and args.profile_kernels > 0
    for e in engs:
        e.profile_enable(0 if solo else args.profile_kernels)
        e.profile_read()  # reset accumulators
    t0 = time.perf_counter()
    for i in range(K):
        if solo and i % args.profile_kernels == 0:
            sync_engines()
            e = engs[i % len(engs)]
            e.profile_enable(1)
            step(i)
            e.sync()
            e.profile_enable(0)
        else:
            step(i)
    if len(engs) > 1:
        sync_engines()
    # K7 + C1: compact detected records, gather them to rank 0 (the only collective)
    n_kept = eng.compact_device(rec.data_ptr(), total * T, kept.data_ptr())
    gathered = (parallel.gather_records(kept[:n_kept], world, rank, dev, force=use_dist)
                if use_dist else kept[:n_kept])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = {}
    for e in engs:
        for k, (ms, cnt) in e.profile_read().items():
            prof[k] = (prof.get(k, (0.0, 0))[0] + ms, prof.get(k, (0.0, 0))[1] + cnt)
        e.profile_enable(0)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        blocks_total = world * K * B
        value = blocks_total / dt
        bytes_per_block = 2 * N_BLOCK + 64 * T
        if pnum:   # the fused kernel is timed in k_correlate's event slot
            prof = {("k_preshift" if k == "k_correlate" else k): v for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k][0])
        dom_ms, dom_cnt = prof[dom]
        if dom_cnt == 0:  # --profile-kernels 0: fall back to the whole step
            dom, avg_ms = "all kernels of one step", dt / K * 1e3
        else:
            avg_ms = dom_ms / dom_cnt
        achieved = bytes_per_block * B / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("bytes_per_launch_at_batch", {}).get(str(B))
            except Exception:
                traffic = None
        line = {
            "metric": "IQ blocks/sec (16384-sample, 1024-chip template)",
            "value": value, "unit": "blocks/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: block_len=16384 history=4096 "
                                   "1023-chip Gold template (10-bit, 1 sample/chip), %s mix, "
                                   "%d blocks per GPU resident in HBM as u8 IQ" % (args.mix, total),
                       "variant": args.variant if not pnum else "preshift(num=%d)" % pnum,
                       "blocks_per_step_per_gpu": B, "templates": T,
                       "carrier_window": [7, 110], "thresholds": "15*snr",
                       "parallelism": "block-shard x%d" % world,
                       "handles_per_gpu": len(engs),
                       "detections_gathered": int(gathered.shape[0])},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": dom_cnt,
                         "algorithmic_bytes_per_launch": bytes_per_block * B,
                         "all_kernels_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items()},
                         # the kernel is VALU/LDS-bound, so the honest secondary view:
                         # nominal 5 N log2 N flop per transform done by THIS kernel (+ 6N per
                         # pointwise product) against the fp32 vector peak
                         "compute": _compute_view(dom, T, B, avg_ms)},
            "data_gen_s": t_gen,
        }
        if world == 1 and args.cpu_seconds > 0:
            ns = min(total, 16384)
            line["cpu_baseline"] = cpu_baseline(
                data[:ns].cpu().numpy(), np.arange(first, first + ns), tpls[0], args.cpu_seconds,
                rec.view(total, T, 64)[:ns, 0].cpu().numpy().view(F.RECORD_DTYPE).reshape(-1), T, pnum)
            if args.cpu_procs > 0 and not pnum:
                line["cpu_baseline"]["all_cores"] = cpu_all_cores(
                    data[:ns].cpu().numpy(), tpls[0], args.cpu_procs)
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
