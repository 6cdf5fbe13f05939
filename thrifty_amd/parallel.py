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

import os
import socket
import subprocess
import sys

import numpy as np

RECORD_BYTES = 64


def shard_range(n_blocks, rank, world):
    """Contiguous [lo, hi) of block positions owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(int(n_blocks), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# What a rank's environment must hold before torch / the HIP runtime initialise, whichever way the
# rank was started (the driver's `python -m torch.distributed.run ... bench.py --gpus N`, our own
# relaunch_under_torchrun(), a user's torchrun): the host driver of these boxes only supports
# dmabuf IPC, and without this RCCL's cross-process buffer sharing fails with
# `hipIpcGetMemHandle: invalid argument`.
RANK_ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def rank_env(environ=None):
    """Apply RANK_ENV (defaults: a value the caller exported wins) to `environ` (default: this
    process) and return what is in effect.  Call it before `import torch` / the first HIP call."""
    env = os.environ if environ is None else environ
    for key, val in RANK_ENV.items():
        env.setdefault(key, val)
    return {key: env.get(key) for key in RANK_ENV}


def cpu_budget():
    """CPUs this process may keep busy: its affinity mask, capped by the cgroup-v2 quota
    (`cpu.max`) of the container -- a box that shows 256 logical CPUs may grant 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def populate_threads(world):
    """Page-table populator threads of a rank's input window (thr_input_window_ex): the CPUs of the
    box divided among the ranks, minus the rank's own host thread, its text formatter and the
    page-locking worker; between 1 and 3 (three is as good as any on an otherwise idle host,
    DESIGN.md section 6; eight ranks on a 16-CPU cgroup get one each)."""
    return max(1, min(3, cpu_budget() // max(1, int(world)) - 3))


_gather_method = "gather"      # or "all_gather" / "host": chosen by gather_selftest() on the real backend
_host_group = None             # method "host": a gloo group beside the device backend's (created collectively)


def gather_records(local, world, rank, device=None, group=None, force=False, method=None):
    """Gather variable-length uint8 [n_i, 64] record tensors to rank 0, in rank order.

    Returns the concatenation on rank 0 and an empty [0, 64] tensor elsewhere.
    Because each rank owns an increasing block range and its records are in
    block order, the result is globally ordered by block index.

    method "gather" (default): counts by all_gather, then ONE padded `dist.gather` to rank 0;
    "all_gather": the padded buffers go to every rank and the others drop them -- the same bytes
    over the most exercised collective, what gather_selftest() falls back to if the backend's
    gather does not work; "host": the same exchange with CPU tensors over a gloo group beside the
    device backend's -- 64 bytes per detection do not need the fabric -- the last resort if neither
    device collective delivers the records."""
    import torch
    import torch.distributed as dist

    if world == 1 and not force:
        return local
    method = method or _gather_method
    dev = local.device if device is None else device
    if method == "host":
        got = gather_records(local.cpu(), world, rank, torch.device("cpu"), group=_host_group, force=force,
                             method="gather")
        return got.to(dev)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    padded = torch.zeros((m, RECORD_BYTES), dtype=torch.uint8, device=dev)
    padded[:local.shape[0]] = local
    if method == "all_gather":
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded, group=group)
        if rank != 0:
            return padded[:0]
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    if rank == 0:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, bufs, dst=0, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(padded, None, dst=0, group=group)
    return padded[:0]


def selftest_counts(world):
    """Record counts of the pre-flight gather: uneven, and one rank with nothing (world > 1)."""
    return [0 if (world > 1 and r == world - 1) else 3 + 2 * (r % 3) for r in range(world)]


def gather_selftest(world, rank, device, group=None):
    """One tiny gather_records() round on the REAL backend, before anything is timed: every rank
    sends selftest_counts()[rank] records stamped (rank, position); rank 0 checks count, order and
    bytes.  If the backend's `gather` raises (an operation it does not offer raises on every rank)
    or delivers something else to rank 0, every rank switches to the all_gather form (decided
    collectively: an all-reduce of the ranks' verdicts; re-checked); if that fails too the run ends
    here with a sentence instead of inside the timed region.  (A collective that ONE rank never
    enters hangs its peers on any backend: that is what the process group's timeout is for, and
    the driver's clock would show it in front of the first step, not inside one.)  -> the method
    in use."""
    import torch
    import torch.distributed as dist
    global _gather_method

    counts = selftest_counts(world)

    def payload(r):
        a = np.zeros((counts[r], RECORD_BYTES), dtype=np.uint8)
        a[:, 0] = r & 0xFF
        a[:, 1] = np.arange(counts[r]) & 0xFF
        a[:, 2:] = (np.arange(2, RECORD_BYTES) * (r + 1)) & 0xFF
        return a

    def one_round(method):
        bad, why = 0, ""
        try:
            got = gather_records(torch.from_numpy(payload(rank)).to(device), world, rank, device, group=group,
                                 force=True, method=method)
            if rank == 0:
                want = np.concatenate([payload(r) for r in range(world)])
                if got.shape != tuple(want.shape) or not np.array_equal(got.cpu().numpy(), want):
                    bad, why = 1, "rank 0 received %s records, expected %s" % (tuple(got.shape), want.shape)
            elif got.shape[0] != 0:
                bad, why = 1, "rank %d kept %d records" % (rank, got.shape[0])
        except Exception as exc:      # (a backend without gather, a refused buffer: say so and fall back)
            bad, why = 1, "%s: %s" % (type(exc).__name__, exc)
        flag = torch.tensor([bad], dtype=torch.int64, device=device)
        dist.all_reduce(flag, group=group)
        return int(flag.item()), why

    failed, why = one_round("gather")
    if failed:
        print("pre-flight: gather over %s failed on %d rank(s)%s; falling back to all_gather"
              % (dist.get_backend(group), failed, " (rank %d: %s)" % (rank, why) if why else ""), file=sys.stderr)
        failed2, why2 = one_round("all_gather")
        _gather_method = "all_gather"
        if failed2:
            print("pre-flight: all_gather over %s failed too%s; the records travel over a gloo group on the host"
                  % (dist.get_backend(group), " (rank %d: %s)" % (rank, why2) if why2 else ""), file=sys.stderr)
            global _host_group
            _host_group = dist.new_group(backend="gloo")      # (collective: every rank is here)
            failed3, why3 = one_round("host")
            if failed3:
                raise SystemExit("pre-flight: neither gather nor all_gather over %s nor a gloo group delivers the "
                                 "ranks' records (rank %d: %s)" % (dist.get_backend(group), rank, why3 or why2 or why))
            _gather_method = "host"
    else:
        _gather_method = "gather"
    return _gather_method


# ---------------------------------------------------------------------------------------------
# `thrifty detect --gpus N`: one process per GPU, each over its contiguous range of the input
# ---------------------------------------------------------------------------------------------
def torchrun_env():
    """(rank, world, local_rank) when started by torch.distributed.run, else (0, None, 0)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]),
                int(os.environ.get("LOCAL_RANK", os.environ["RANK"])))
    return 0, None, 0


def sharded_env(gpus=1):
    """(rank, world, local_rank) of a rank of a sharded `--gpus N` run, else (0, None, 0).

    A rank is a process that `relaunch_under_torchrun` started (it sets THRIFTY_SHARDED=1) -- or one
    that the user started with torch.distributed.run themselves (`torchrun ... -m
    thrifty_amd.detect --gpus N`: torchrun's own TORCHELASTIC_RUN_ID is set and WORLD_SIZE == N):
    such a process must never re-launch, or N ranks would start N x N processes writing one file.
    RANK / WORLD_SIZE of a different size under `--gpus N` is refused; without `--gpus` whatever
    RANK / WORLD_SIZE an unrelated launcher, MPI wrapper or container job left in the
    environment does not turn a plain `thrifty detect` into a rank."""
    if os.environ.get("THRIFTY_SHARDED") == "1":
        return torchrun_env()
    rank, world, local = torchrun_env()
    if gpus > 1 and world is not None and "TORCHELASTIC_RUN_ID" in os.environ:
        if world != gpus:
            raise SystemExit("--gpus %d inside a torch.distributed.run job of %d ranks: start one rank "
                             "per GPU (--nproc-per-node %d) or drop the launcher and let --gpus start "
                             "the ranks" % (gpus, world, gpus))
        return rank, world, local
    return 0, None, 0


def peek_gpus(argv):
    """Value of --gpus in argv (1 if absent) without parsing anything else."""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(gpus, argv):
    """Start `gpus` ranks of the CLI that is running now (same module, same arguments) on this
    node; returns the launcher's exit status."""
    import __main__
    spec = getattr(__main__, "__spec__", None)
    target = ["-m", spec.name] if spec is not None and spec.name else [os.path.abspath(sys.argv[0])]
    env = dict(os.environ)
    rank_env(env)                                       # dmabuf IPC (RCCL across processes)
    env["THRIFTY_SHARDED"] = "1"                        # marks the children as ranks of THIS CLI
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + target + list(argv)
    return subprocess.call(cmd, env=env)


_rank_ready = {}      # (backend, local) -> device, once init_rank has run in this process


def init_rank(rank, world, local, backend="nccl"):
    """Everything a rank needs of torch BEFORE it touches the engine: `import torch` (which loads and
    registers thousands of device code objects), the process group on `backend`, and the rehearsal of
    the record gather -> the torch device the collectives run on.  `thrifty detect --gpus N` calls this
    before it constructs the Detector: an engine with an input window has library threads inside the
    HIP runtime (page-locking the mapping) from the moment it exists, and loading torch / starting
    RCCL beside them is a way to find out which of the runtime's locks they share.  Idempotent;
    run_sharded calls it again and gets the same device."""
    import torch
    import torch.distributed as dist
    key = (backend, local)
    if key in _rank_ready and dist.is_initialized():
        return _rank_ready[key]
    if backend == "nccl":       # RCCL; "gloo" (CPU tensors) is for the tests and the rehearsals
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=dev)      # (default timeout: a rank's shard may be hours of capture)
    else:
        dev = torch.device("cpu")
        if not dist.is_initialized():
            dist.init_process_group(backend)
    # before any work: the record gather must run on this backend with uneven and empty ranks
    # (falls back to the all_gather form, or ends the run with a sentence)
    gather_selftest(world, rank, dev)
    _rank_ready[key] = dev
    return dev


def run_sharded(detections, rank, world, local, output_file, backend="nccl"):
    """Body of one rank of `thrifty detect --gpus N`: run this rank's block range, gather the
    detected records to rank 0 (RCCL), write one .toad there in input order.

    `detections` is a Detector over a reader already restricted with `.shard(rank, world)`.
    Each record's timestamp travels in its `reserved` field.  If the reference would have
    raised IndexError on some block (carrier_sync.py:187), rank 0 writes the detections before
    that block and every rank raises, like the single-process loop."""
    import torch
    import torch.distributed as dist
    from thrifty_amd import _native
    from thrifty_amd.detect import _offset_mode

    dev = init_rank(rank, world, local, backend)
    # this rank's detected records, each with its timestamp in `reserved` (a mapped shard runs
    # inside the library: thr_run_card / thr_run_stream with a record sink)
    error, mine = None, np.zeros(0, dtype=_native.RECORD_DTYPE)
    if hasattr(detections, "detected_records") and getattr(detections, "_library_loop_ready", lambda: False)():
        sink = []
        try:
            mine = detections._run_library_loop(want_records=True, partial=sink)[1]
        except IndexError as exc:
            error = exc
            mine = np.concatenate(sink) if sink else mine
    else:
        chunks = []
        try:
            for stamps, recs in detections.iter_detected_records():
                recs = recs.copy()
                recs["reserved"] = np.ascontiguousarray(stamps, dtype=np.float64).view(np.uint64)
                chunks.append(recs)
        except IndexError as exc:
            error = exc
        mine = np.concatenate(chunks) if chunks else mine
    failed = torch.tensor([1 if error is not None else 0], dtype=torch.int64, device=dev)
    flags = [torch.zeros_like(failed) for _ in range(world)]
    dist.all_gather(flags, failed)
    flags = [int(f.item()) for f in flags]
    first_bad = flags.index(1) if 1 in flags else world
    if rank > first_bad:
        mine = mine[:0]                 # the single-process loop never got this far
    local_t = torch.from_numpy(mine.view(np.uint8).reshape(-1, RECORD_BYTES).copy()).to(dev)
    gathered = gather_records(local_t, world, rank, dev, force=True)
    if rank == 0 and output_file is not None:
        recs = gathered.cpu().numpy().reshape(-1).view(_native.RECORD_DTYPE)
        stamps = recs["reserved"].view(np.float64)
        step = 1 << 16
        for s in range(0, len(recs), step):
            # (multi-template detection: records arrive ordered [block][template]; the txid column
            # is the template id, as the single-process writer prints it)
            text = _native.format_toad(
                recs[s:s + step], stamps[s:s + step], detections.new_len, rxid=detections.rxid,
                with_txid=getattr(detections, "_multi", False),
                carrier_offset_f32=_offset_mode(getattr(detections, "_offset_type", float)))
            output_file.write(text.decode("ascii"))
        output_file.flush()
    dist.barrier()
    dist.destroy_process_group()
    _rank_ready.clear()
    if first_bad < world:
        raise error if error is not None else IndexError(
            "rank %d hit a block on which the reference raises IndexError" % first_bad)
