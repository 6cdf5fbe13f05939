"""Dev: run ONE rank of `thrifty detect --gpus 1` (RCCL, world size 1) over and over and, when a run
does not come back, say where it sits: the kernel wait channel of every thread and (SIGABRT under
-X faulthandler) the Python stack of every thread.

    python scripts/hang_probe.py [runs] [seconds per run] [direct|torchrun|window_first|torch_first]

window_first / torch_first: scripts/hang_probe_child.py instead of the CLI -- an engine with an input
window over a 3 GiB mapping and `import torch` + RCCL start-up, in that order or the other.
"""
import glob
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from thrifty_amd import block_data

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 60
mode = sys.argv[3] if len(sys.argv) > 3 else "direct"
g = np.load(os.path.join(ROOT, "tests", "golden", "c2.npz"))
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, "template.npy"), g["template"])
open(os.path.join(tmp, "detector.cfg"), "w").write(
    "rxid: 0\nsample_rate: 2.4M\nblock_size: 16384\nblock_history: 4096\n"
    "carrier_window: 7 - 110\ncarrier_threshold: 15 * snr\ncorr_threshold: 15*snr\n"
    "template: %s\n" % os.path.join(tmp, "template.npy"))
open(os.path.join(tmp, "rx.card"), "w").write("# synthetic\n" + "".join(
    block_data.card_line(1000.0 + i, int(g["block_idx"][i]), g["blocks"][i]) for i in range(len(g["blocks"]))))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def threads_of(pid):
    out = []
    for task in sorted(glob.glob("/proc/%d/task/*" % pid)):
        def rd(name):
            try:
                return open(os.path.join(task, name)).read().strip()
            except OSError as exc:
                return "?" + type(exc).__name__
        out.append("  tid %s %-18s wchan %-28s syscall %s" % (os.path.basename(task), rd("comm"), rd("wchan"),
                                                            rd("syscall").split(" ")[0]))
    return "\n".join(out)


def children(pid):
    try:
        return [int(p) for p in subprocess.check_output(["pgrep", "-P", str(pid)]).split()]
    except subprocess.CalledProcessError:
        return []


hung = 0
for i in range(runs):
    port = free_port()
    args = ["-m", "thrifty_amd.detect", "--gpus", "1", os.path.join(tmp, "rx.card"), "--quiet", "-c",
            os.path.join(tmp, "detector.cfg"), "-o", os.path.join(tmp, "rank%d.toad" % i)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, THRIFTY_SHARDED="1",
               PYTHONFAULTHANDLER="1")
    if mode in ("window_first", "torch_first"):
        big = os.path.join(tmp, "big.bin")
        if not os.path.exists(big):
            with open(big, "wb") as f:
                chunk = np.random.default_rng(0).integers(96, 160, 64 << 20, dtype=np.uint8).tobytes()
                for _ in range(48):
                    f.write(chunk)
        env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, "-X", "faulthandler", os.path.join(ROOT, "scripts", "hang_probe_child.py"), mode, big]
    elif mode == "direct":
        env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, "-X", "faulthandler"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    t0 = time.time()
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = p.communicate(timeout=limit)
        print("run %d: rc %d in %.1f s" % (i, p.returncode, time.time() - t0), flush=True)
        if p.returncode != 0:
            print(err[-3000:])
    except subprocess.TimeoutExpired:
        hung += 1
        print("run %d: NOT BACK after %.0f s" % (i, limit), flush=True)
        victims = [p.pid]
        for pid in list(victims):
            victims += children(pid)
        for pid in list(victims[1:]):
            victims += children(pid)
        for pid in victims:
            try:
                print(" pid %d: %s" % (pid, open("/proc/%d/cmdline" % pid).read().replace("\0", " ")[:150]))
            except OSError:
                continue
            print(threads_of(pid))
        for pid in reversed(victims):
            try:
                os.kill(pid, signal.SIGABRT)
            except OSError:
                pass
        try:
            out, err = p.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        print(err[-6000:], flush=True)
        if hung >= 2:
            break
print("hung %d of %d" % (hung, i + 1))
