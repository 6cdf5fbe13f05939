"""One rank of the multi-GPU parity run (started by tests/test_gpu_distributed.py through
`python -m torch.distributed.run`): shard_range -> Engine.detect_device -> compact_device ->
gather_records over the **nccl** (= RCCL) backend; rank 0 saves what it gathered.

All ranks synthesise the same seeded blocks and take their own contiguous range, as
`bench.py --gpus N` and `thrifty detect --gpus N` do."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_blocks(n_blocks, seed, templates=1):
    """-> (template(s), blocks): `templates` > 1 gives [T, W] Gold templates (BASELINE configs[4])
    and blocks whose bursts use them in turn."""
    from thrifty_amd import synth
    from thrifty_amd.detect import unique_window
    tpls = np.stack([synth.gold_template(10, 2 + i) for i in range(templates)]).astype(np.float64)
    rng = np.random.default_rng(seed)
    win = unique_window(16384, 4096, tpls.shape[1])
    blocks = np.concatenate([
        synth.synth_blocks(rng, len(part), 16384, tpls[t], win, signal_frac=0.6)[0]
        for t, part in enumerate(np.array_split(np.arange(n_blocks), templates)) if len(part)])
    blocks = blocks[rng.permutation(n_blocks)]
    return (tpls if templates > 1 else tpls[0]), blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=203)
    ap.add_argument("--seed", type=int, default=77)
    ap.add_argument("--out", required=True)
    ap.add_argument("--templates", type=int, default=1,
                    help="T > 1: BASELINE configs[4] -- T templates per block, records [block][template]")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: every rank uses GPU 0 and the records travel as CPU tensors -- the "
                         "sharding / ordering logic of world sizes a 1-GPU box cannot give to RCCL")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from thrifty_amd import _native as F
    from thrifty_amd import parallel

    rank, world, local = parallel.torchrun_env()
    assert world is not None, "start me with torch.distributed.run"
    if args.backend == "gloo":
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    assert dist.get_backend() == args.backend and dist.get_world_size() == world
    tpl, blocks = make_blocks(args.blocks, args.seed, args.templates)
    T = args.templates
    lo, hi = parallel.shard_range(args.blocks, rank, world)
    n = hi - lo
    eng = F.Engine(16384, 4096, tpl, (0, 15, 0), (7, 110), (0, 15, 0), device_id=local, max_batch=max(n, 1))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    data = torch.from_numpy(blocks[lo:hi].copy()).to(dev)
    idx = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    rec = torch.zeros((max(n, 1) * T, 64), dtype=torch.uint8, device=dev)
    kept = torch.zeros_like(rec)
    # (the engine runs on its own non-blocking stream: torch's fills on the null stream must have
    # landed before it writes records into these buffers)
    torch.cuda.synchronize()
    if n:
        eng.detect_device(data.data_ptr(), F.THR_IN_U8, n, rec.data_ptr(), idx.data_ptr())
    n_kept = eng.compact_device(rec.data_ptr(), n * T, kept.data_ptr())
    if args.backend == "nccl":
        gathered = parallel.gather_records(kept[:n_kept], world, rank, dev, force=True)
    else:
        gathered = parallel.gather_records(kept[:n_kept].cpu(), world, rank, torch.device("cpu"), force=True)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(args.out, gathered.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
