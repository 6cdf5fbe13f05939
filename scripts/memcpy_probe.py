"""Dev: does hipMemcpyAsync from PAGEABLE host memory return before the copy is done?  (decides what
thr_submit*() may promise about its input arrays; see include/thrifty_hip.h)"""
import ctypes as C, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from thrifty_amd import _native as F
F.load_library()
hip = C.CDLL(None)      # the HIP runtime the engine library already brought in (torch's copy, RTLD_GLOBAL)
nbytes = 64 << 20
src = np.random.default_rng(0).integers(0, 255, nbytes, dtype=np.uint8)
dst = C.c_void_p(); st = C.c_void_p()
assert hip.hipMalloc(C.byref(dst), C.c_size_t(nbytes)) == 0
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
for rep in range(4):
    t0 = time.perf_counter()
    assert hip.hipMemcpyAsync(dst, C.c_void_p(src.ctypes.data), C.c_size_t(nbytes), 1, st) == 0
    t1 = time.perf_counter()
    assert hip.hipStreamSynchronize(st) == 0
    t2 = time.perf_counter()
    print("pageable 64 MiB: call returned after %.2f ms, stream idle after %.2f ms more" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
