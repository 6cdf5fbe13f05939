"""world_size-2 CPU test (gloo) of the only collective on the path: the gather of
detection records to rank 0, plus the block-range sharding."""
import os
import socket

import numpy as np
import pytest

from thrifty_amd import parallel


def test_shard_range_covers_everything():
    for n, w in [(10, 2), (10, 3), (7, 8), (1 << 20, 8), (0, 4)]:
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, counts, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = counts[rank]
        local = torch.zeros((n, 64), dtype=torch.uint8)
        # block_idx field (first 8 bytes) = global position, as a shard would produce
        lo = sum(counts[:rank])
        idx = torch.arange(lo, lo + n, dtype=torch.int64)
        local[:, :8] = idx.view(torch.uint8).view(n, 8) if n else local[:, :8]
        local[:, 8] = rank + 1
        out = parallel.gather_records(local, world, rank)
        if rank == 0:
            ret.put(out.numpy().copy())
        else:
            assert out.shape == (0, 64)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(5, 3), (0, 4), (7, 0)])
def test_gather_records_two_ranks(counts):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (sum(counts), 64)
    idx = got[:, :8].copy().view(np.int64).reshape(-1)
    assert np.array_equal(idx, np.arange(sum(counts)))          # globally ordered
    assert list(got[:, 8]) == [1] * counts[0] + [2] * counts[1]  # rank order
