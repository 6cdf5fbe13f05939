"""Block-shard data parallelism: one process per GPU, no data-path collective.

IQ blocks are independent once they carry their own `history_len` overlap
(reference card_reader.c:69-75 / block_data.py:93-98), so rank r simply owns a
contiguous range of block indices and runs the whole hot path on it.  The only
exchange is C1: the gather of the (already compacted, already ordered) 64-byte
detection records to rank 0 -- RCCL over xGMI on GPUs (backend "nccl"), gloo
on CPU in the tests.  Payload is KBs-MBs, i.e. latency bound; do it once per
large batch, never per launch.
"""
from __future__ import annotations

RECORD_BYTES = 64


def shard_range(n_blocks, rank, world):
    """Contiguous [lo, hi) of block positions owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(int(n_blocks), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(local, world, rank, device=None, group=None, force=False):
    """Gather variable-length uint8 [n_i, 64] record tensors to rank 0, in rank order.

    Returns the concatenation on rank 0 and an empty [0, 64] tensor elsewhere.
    Because each rank owns an increasing block range and its records are in
    block order, the result is globally ordered by block index.
    """
    import torch
    import torch.distributed as dist

    if world == 1 and not force:
        return local
    dev = local.device if device is None else device
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    padded = torch.zeros((m, RECORD_BYTES), dtype=torch.uint8, device=dev)
    padded[:local.shape[0]] = local
    if rank == 0:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, bufs, dst=0, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(padded, None, dst=0, group=group)
    return padded[:0]
