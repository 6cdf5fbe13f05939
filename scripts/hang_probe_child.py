"""Dev (child of hang_probe.py's `order` mode): one process that starts an engine with an input
window over a large mapping and loads torch + starts RCCL -- in the given order.

    python scripts/hang_probe_child.py window_first|torch_first <file>
"""
import mmap
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from thrifty_amd import _native as F

order, path = sys.argv[1], sys.argv[2]
tpl = np.load(os.path.join(ROOT, "tests", "golden", "c2.npz"))["template"]


def window():
    eng = F.Engine(16384, 4096, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=2048)
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    eng.input_window(memoryview(mm), populate_threads=3)
    return eng, mm, f


def torch_up():
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    x = torch.ones(4, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    return dist


t0 = time.time()
if order == "window_first":
    keep = window()
    d = torch_up()
else:
    d = torch_up()
    keep = window()
# read a little through the window, as a run would
eng, mm, f = keep
view = memoryview(mm)
rec = np.zeros(1 << 16, dtype=F.RECORD_DTYPE)
st = eng.run_stream(view[:64 << 20], first_block_idx=0, rec_out=rec)
d.barrier()
d.destroy_process_group()
eng.input_window(None)
print("%s ok in %.1f s, %d blocks" % (order, time.time() - t0, st["blocks"]))
