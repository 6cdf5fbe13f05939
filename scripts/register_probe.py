"""Dev: is hipHostRegister on an mmap-ed input file worth it?  Times, for a file in the page cache:
  (a) hipMemcpyAsync of 64 MiB chunks straight from the read-only mapping (pageable: the HIP runtime
      stages them) -- what thr_submit_card / thr_submit_stream do today;
  (b) hipHostRegister of the mapping (whole file at once, and chunk by chunk), then the same copies
      (page-locked in place: DMA straight from the page cache, no staging copy).
Result recorded in profiles/README.md."""
import ctypes as C
import mmap
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import _native as F

F.load_library()
hip = C.CDLL(None)
hip.hipGetErrorString.restype = C.c_char_p
total = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 2048 << 20
chunk = 64 << 20
path = os.path.join(sys.argv[2] if len(sys.argv) > 2 else tempfile.gettempdir(), "register_probe.bin")
with open(path, "wb") as f:
    blk = np.random.default_rng(0).integers(0, 255, chunk, dtype=np.uint8).tobytes()
    for _ in range(total // chunk):
        f.write(blk)
dst = C.c_void_p()
st = C.c_void_p()
assert hip.hipMalloc(C.byref(dst), C.c_size_t(2 * chunk)) == 0
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0


def copy_all(base, label):
    t0 = time.perf_counter()
    for i in range(total // chunk):
        rc = hip.hipMemcpyAsync(C.c_void_p(dst.value + (i & 1) * chunk), C.c_void_p(base + i * chunk),
                                C.c_size_t(chunk), 1, st)
        assert rc == 0, hip.hipGetErrorString(rc)
    t1 = time.perf_counter()
    assert hip.hipStreamSynchronize(st) == 0
    t2 = time.perf_counter()
    print("%-46s calls %.1f ms, done %.1f ms -> %.1f GB/s" % (label, (t1 - t0) * 1e3, (t2 - t0) * 1e3,
                                                            total / (t2 - t0) / 1e9))


for access, prot_name in ((mmap.ACCESS_READ, "PROT_READ shared"), (mmap.ACCESS_COPY, "private copy-on-write")):
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=access)
    arr = np.frombuffer(mm, dtype=np.uint8)
    base = arr.ctypes.data
    t0 = time.perf_counter()
    s = int(arr[::4096].sum())               # fault every page in (page cache -> this mapping)
    print("[%s] first touch of %d MiB: %.1f ms" % (prot_name, total >> 20, (time.perf_counter() - t0) * 1e3))
    copy_all(base, "pageable, warm-up")
    copy_all(base, "pageable")
    for flags, fname in ((0, "hipHostRegisterDefault"), (8, "hipHostRegisterReadOnly")):
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(C.c_void_p(base), C.c_size_t(total), C.c_uint(flags))
        dt = time.perf_counter() - t0
        if rc != 0:
            print("hipHostRegister(%s) of the whole mapping: FAILED (%s) after %.1f ms" % (
                fname, hip.hipGetErrorString(rc).decode(), dt * 1e3))
            hip.hipGetLastError()
            continue
        print("hipHostRegister(%s) of the whole mapping: %.1f ms (%.1f GB/s of pinning)" % (
            fname, dt * 1e3, total / dt / 1e9))
        copy_all(base, "registered")
        copy_all(base, "registered (again)")
        t0 = time.perf_counter()
        assert hip.hipHostUnregister(C.c_void_p(base)) == 0
        print("hipHostUnregister: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        # chunk-wise: register chunk i + 1 while chunk i copies (what a streaming reader could do)
        t0 = time.perf_counter()
        ok = True
        for i in range(total // chunk):
            rc = hip.hipHostRegister(C.c_void_p(base + i * chunk), C.c_size_t(chunk), C.c_uint(flags))
            if rc != 0:
                ok = False
                break
            hip.hipMemcpyAsync(C.c_void_p(dst.value + (i & 1) * chunk), C.c_void_p(base + i * chunk),
                               C.c_size_t(chunk), 1, st)
        hip.hipStreamSynchronize(st)
        dt = time.perf_counter() - t0
        for j in range(i + (1 if ok else 0)):
            hip.hipHostUnregister(C.c_void_p(base + j * chunk))
        print("chunk-wise register + copy (%s): %s, %.1f ms -> %.1f GB/s" % (
            fname, "ok" if ok else "FAILED at chunk %d" % i, dt * 1e3, total / dt / 1e9))
        break
    del arr
    mm.close()
os.unlink(path)
