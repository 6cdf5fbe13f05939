"""TEST INFRASTRUCTURE ONLY -- ctypes access to oracle/_ref/libfastcard_readers.so: the
reference's OWN native block readers (fastcard/card_reader.c + lib/base64.c, fastcard/raw_reader.c),
compiled by oracle/Makefile from the sources under /root/reference (nothing copied).  Used by the
tests to pin the host framing / base64 decode of `thrifty_amd.block_data` and the engine's
`.card` ingest against the reference's native twin of `card_reader` / `block_reader`
(SURVEY.md 8(a) a2, 8(f) rank 1).  Never imported by the product.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libfastcard_readers.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(LIB_PATH)
        for name in ("ref_card_open", "ref_raw_open"):
            fn = getattr(lib, name)
            fn.restype = C.c_void_p
            fn.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p]
        lib.ref_next.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long),
                                 C.POINTER(C.c_longlong)]
        lib.ref_close.argtypes = [C.c_void_p]
        lib.ref_close.restype = None
        _lib = lib
    return _lib


def read_blocks(path, block_size, history_size, card, initial=None, max_blocks=1 << 30):
    """Run the reference reader over `path` -> (list of (tv_sec, tv_usec, index, bytes uint8[2 N]),
    final return code: 1 = clean end of file, < 0 = the reader's error code)."""
    lib = _load()
    init = None
    if initial is not None:
        init = np.ascontiguousarray(initial, dtype=np.uint8)
        assert init.size == 2 * block_size
    h = (lib.ref_card_open if card else lib.ref_raw_open)(
        os.fsencode(path), block_size, history_size, None if init is None else init.ctypes.data)
    if not h:
        raise OSError("reference reader could not open %s" % path)
    out = []
    buf = np.zeros(2 * block_size, dtype=np.uint8)
    sec, usec, idx = C.c_long(), C.c_long(), C.c_longlong()
    rc = 0
    try:
        while len(out) < max_blocks:
            rc = lib.ref_next(h, buf.ctypes.data, C.byref(sec), C.byref(usec), C.byref(idx))
            if rc != 0:
                break
            out.append((sec.value, usec.value, idx.value, buf.copy()))
    finally:
        lib.ref_close(h)
    return out, rc
