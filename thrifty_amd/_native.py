"""ctypes binding of libthriftyhip.so (include/thrifty_hip.h).

There is deliberately no fallback: if the shared library is missing or no
MI355X is visible, constructing an :class:`Engine` raises.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("THRIFTY_HIP_LIB") or os.path.join(HERE, "libthriftyhip.so")  # env: A/B builds

THR_IN_U8 = 0
THR_IN_C64 = 1
FLAG_CARRIER = 1
FLAG_CORR = 2
FLAG_INDEX_ERROR = 4
FLAG_INT_OFFSET = 8
FLAG_FIT_UNCONVERGED = 16
N_KERNEL_SLOTS = 5

ABI_VERSION = 9     # THR_ABI_VERSION of include/thrifty_hip.h

EXPORTS = [
    "thr_abi_version", "thr_last_error", "thr_create", "thr_destroy", "thr_detect",
    "thr_create_preshift", "thr_create_fastdet", "thr_create_ex", "thr_plan_sections", "thr_host_register", "thr_host_unregister", "thr_input_window", "thr_detect_card", "thr_detect_stream", "thr_detect_stream_device", "thr_detect_device", "thr_sync", "thr_set_stream", "thr_compact_device",
    "thr_profile_enable", "thr_profile_read", "thr_kernel_name", "thr_debug_fft",
    "thr_debug_stage", "thr_debug_stage_offsets", "thr_identify", "thr_frame_card",
    "thr_submit", "thr_submit_card", "thr_submit_stream", "thr_collect", "thr_inputs_consumed", "thr_poll",
    "thr_set_stream_default", "thr_format_toad",
    "thr_run_card", "thr_run_stream", "thr_get_settings", "thr_input_window_ex", "thr_input_window_release", "thr_detect_offsets", "thr_set_wait_mode", "thr_debug_window", "thr_debug_window_times", "thr_debug_correlate_geom", "thr_debug_sections", "thr_debug_pipe_times", "thr_get_path_info",
]
ERR_ARG, ERR_DEVICE, ERR_STATE, ERR_INDEX = -1, -2, -3, -4       # THR_ERR_*
VARIANT_DEFAULT, VARIANT_PRESHIFT, VARIANT_FASTDET = 0, 1, 2      # THR_VARIANT_*
INTERPOLATORS = {"parabolic": 0, "none": 1, "gaussian": 2, "cosine": 3}      # THR_INTERP_*
PATHS = {"auto": 0, "multipass": 1, "unsectioned": 2, "generic_rows": 3, "unsectioned_generic_rows": 4}      # THR_PATH_*
MAX_IN_FLIGHT = 3       # THR_MAX_IN_FLIGHT
TOAD_LINE_MAX = 384     # THR_TOAD_LINE_MAX


class ThrSettings(C.Structure):
    _fields_ = [
        ("block_len", C.c_int32), ("history_len", C.c_int32),
        ("n_templates", C.c_int32), ("template_len", C.c_int32),
        ("templates", C.POINTER(C.c_double)),
        ("carrier_len", C.c_int32), ("carrier_window", C.c_int32 * 2),
        ("carrier_thresh", C.c_double * 3), ("corr_thresh", C.c_double * 3),
        ("device_id", C.c_int32), ("max_batch", C.c_int32),
    ]


class ThrPathInfo(C.Structure):
    _fields_ = [
        ("n_sections", C.c_int32), ("section_len", C.c_int32), ("rows_lo", C.c_int32), ("rows_hi", C.c_int32),
        ("why_unsectioned", C.c_int32), ("n_templates", C.c_int32),
        ("carrier_kernel", C.c_char * 48), ("correlate_kernel", C.c_char * 48), ("text", C.c_char * 256),
    ]


# THR_WHY_*
WHY_UNSECTIONED = ("sectioned", "path", "variant", "stddev", "geometry", "block_len")


class ThrRunOpts(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32), ("batch_blocks", C.c_int32), ("out_fd", C.c_int32),
        ("with_rxid", C.c_int32), ("rxid", C.c_int64), ("with_txid", C.c_int32),
        ("carrier_offset_mode", C.c_int32), ("timestamp", C.c_double),
        ("rec_out", C.c_void_p), ("rec_capacity", C.c_size_t),
    ]


class ThrRunStats(C.Structure):
    _fields_ = [
        ("blocks", C.c_uint64), ("detections", C.c_uint64), ("batches", C.c_uint64),
        ("bytes_in", C.c_uint64), ("text_bytes", C.c_uint64), ("index_error_at", C.c_uint64),
        ("index_error_block", C.c_int64), ("index_error_bin", C.c_int32), ("reserved_", C.c_int32),
        ("total_s", C.c_double), ("frame_s", C.c_double), ("submit_s", C.c_double), ("wait_s", C.c_double),
        ("format_s", C.c_double), ("write_s", C.c_double),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved_"}


# numpy mirror of thr_record (64 bytes)
RECORD_DTYPE = np.dtype([
    ("block_idx", "<i8"), ("flags", "<u4"), ("template_id", "<i4"),
    ("carrier_bin", "<i4"), ("corr_sample", "<i4"),
    ("carrier_offset", "<f8"), ("corr_offset", "<f8"),
    ("carrier_energy", "<f4"), ("carrier_noise", "<f4"),
    ("corr_energy", "<f4"), ("corr_noise", "<f4"), ("reserved", "<u8"),
])
assert RECORD_DTYPE.itemsize == 64


class NativeError(RuntimeError):
    """An error status of the library: args = (message[, status code[, thr_run_* statistics]])."""

    def __str__(self):
        return str(self.args[0]) if self.args else ""

    @property
    def code(self):
        return self.args[1] if len(self.args) > 1 else None


_lib = None


def _share_hip_runtime_with_torch():
    """A process can hold only ONE HIP runtime.  PyTorch-ROCm wheels bundle their
    own libamdhip64 (SONAME libamdhip64.so.7, same as /opt/rocm's); if both copies
    get loaded, whichever initialises second sees "No HIP GPUs".  bench.py and the
    multi-GPU gather need torch in the same process, so when a torch wheel with a
    bundled runtime is installed we load *that* copy first (RTLD_GLOBAL); our
    DT_NEEDED libamdhip64.so.7 then resolves to it by SONAME.  Without torch the
    system ROCm runtime is used.  THRIFTY_HIP_RUNTIME=system skips this."""
    if os.environ.get("THRIFTY_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load_library():
    """dlopen the engine; raises NativeError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "%s not found: build it with `python -m thrifty_amd.build` (needs hipcc). "
            "thrifty_amd has no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    vp, i64p = C.c_void_p, C.POINTER(C.c_int64)
    lib.thr_abi_version.restype = C.c_int
    if lib.thr_abi_version() != ABI_VERSION:
        raise NativeError("%s has ABI version %d, this package needs %d: rebuild it with "
                          "`python -m thrifty_amd.build --force`" % (LIB_PATH, lib.thr_abi_version(), ABI_VERSION))
    lib.thr_last_error.restype = C.c_char_p
    lib.thr_kernel_name.restype = C.c_char_p
    lib.thr_kernel_name.argtypes = [C.c_int]
    lib.thr_create.argtypes = [C.POINTER(ThrSettings), C.POINTER(vp)]
    lib.thr_create_preshift.argtypes = [C.POINTER(ThrSettings), C.c_int, C.POINTER(vp)]
    lib.thr_create_fastdet.argtypes = [C.POINTER(ThrSettings), C.POINTER(vp)]
    lib.thr_input_window.argtypes = [vp, vp, C.c_size_t]
    lib.thr_input_window_ex.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_size_t]
    lib.thr_run_card.argtypes = [vp, vp, C.c_size_t, C.POINTER(ThrRunOpts), C.POINTER(ThrRunStats)]
    lib.thr_run_stream.argtypes = [vp, vp, C.c_size_t, C.c_int64, C.POINTER(ThrRunOpts), C.POINTER(ThrRunStats)]
    lib.thr_get_settings.argtypes = [vp, C.POINTER(ThrSettings)]
    lib.thr_host_register.argtypes = [vp, C.c_size_t]
    lib.thr_host_unregister.argtypes = [vp]
    ip = C.POINTER(C.c_int)
    lib.thr_plan_sections.argtypes = [C.c_int, C.c_int, C.c_int, ip] + [C.c_int * 8] * 5
    lib.thr_create_ex.argtypes = [C.POINTER(ThrSettings), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.thr_destroy.argtypes = [vp]
    lib.thr_destroy.restype = None
    lib.thr_detect.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, vp]
    lib.thr_detect_card.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp]
    lib.thr_detect_offsets.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, vp, vp]
    lib.thr_frame_card.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_size_t, vp, vp, vp,
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.thr_detect_device.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, vp]
    lib.thr_detect_stream.argtypes = [vp, vp, C.c_size_t, C.c_int64, vp, C.c_size_t,
                                      C.POINTER(C.c_size_t)]
    lib.thr_detect_stream_device.argtypes = [vp, vp, vp, C.c_size_t, vp]
    u64p = C.POINTER(C.c_uint64)
    lib.thr_submit.argtypes = [vp, vp, C.c_int, vp, C.c_size_t, vp, u64p]
    lib.thr_submit_card.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp, u64p]
    lib.thr_submit_stream.argtypes = [vp, vp, C.c_size_t, C.c_int64, vp, C.c_size_t,
                                      C.POINTER(C.c_size_t), u64p]
    lib.thr_collect.argtypes = [vp, C.c_uint64]
    lib.thr_inputs_consumed.argtypes = [vp, C.c_uint64]
    lib.thr_poll.argtypes = [vp, C.c_uint64, C.POINTER(C.c_int)]
    lib.thr_set_stream_default.argtypes = [vp]
    lib.thr_format_toad.argtypes = [vp, vp, C.c_size_t, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int,
                                    vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.thr_sync.argtypes = [vp]
    lib.thr_set_stream.argtypes = [vp, vp]
    lib.thr_compact_device.argtypes = [vp, vp, C.c_size_t, vp, C.POINTER(C.c_size_t)]
    lib.thr_profile_enable.argtypes = [vp, C.c_int]
    lib.thr_profile_read.argtypes = [vp, C.POINTER(C.c_double), i64p]
    lib.thr_debug_fft.argtypes = [vp, vp, C.c_int, C.c_size_t, vp]
    lib.thr_debug_stage.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, vp, vp]
    lib.thr_debug_stage_offsets.argtypes = [vp, vp, C.c_int, C.c_size_t, C.c_int, vp, vp, vp]
    lib.thr_identify.argtypes = [C.c_int, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp,
                                 vp, C.POINTER(C.c_size_t)]
    _lib = lib
    return lib


def _check(lib, rc):
    if rc != 0:
        raise NativeError("libthriftyhip: %s (code %d)" % (lib.thr_last_error().decode(), rc))


def frame_card(buf, start, stop, block_len, at_eof, max_records):
    """thr_frame_card over buf[start:stop] (bytes-like: bytearray, mmap, ...) ->
    (timestamps float64[n], block_idx int64[n], payload_off int64[n] relative to buf, next start)."""
    lib = load_library()
    arr = np.frombuffer(buf, dtype=np.uint8)
    cap = int(min(max_records, (stop - start) // 16 + 1))
    ts = np.empty(cap, dtype=np.float64)
    idx = np.empty(cap, dtype=np.int64)
    off = np.empty(cap, dtype=np.int64)
    n, used = C.c_size_t(0), C.c_size_t(0)
    _check(lib, lib.thr_frame_card(arr.ctypes.data + start, stop - start, int(block_len), int(bool(at_eof)),
                                   cap, ts.ctypes.data, idx.ctypes.data, off.ctypes.data,
                                   C.byref(n), C.byref(used)))
    del arr
    k = n.value
    return ts[:k], idx[:k], off[:k] + start, start + used.value


def format_toad(recs, timestamps, new_len, rxid=None, with_txid=False, carrier_offset_f32=False):
    """thr_format_toad: `.toad` text (bytes, one '\\n'-terminated line per record) for a batch of
    DETECTED records -- the text `DetectionResult.serialize()` produces, without per-record
    objects.  with_txid: prepend each record's template_id as the txid column."""
    lib = load_library()
    recs = np.ascontiguousarray(recs, dtype=RECORD_DTYPE).reshape(-1)
    ts = np.ascontiguousarray(timestamps, dtype=np.float64).reshape(-1)
    n = len(recs)
    assert len(ts) == n
    buf = np.empty(max(1, n * TOAD_LINE_MAX), dtype=np.uint8)
    used = C.c_size_t(0)
    _check(lib, lib.thr_format_toad(recs.ctypes.data, ts.ctypes.data, n, int(new_len),
                                    0 if rxid is None else 1, 0 if rxid is None else int(rxid),
                                    int(bool(with_txid)), int(carrier_offset_f32),
                                    buf.ctypes.data, buf.size, C.byref(used)))
    return buf[:used.value].tobytes()


def format_toad_address():
    """Address of thr_format_toad in the loaded library (0: no library) -- handed to
    thrifty_amd._fastresults, which links nothing, so that DetectionResult.serialize() of an engine
    record is the library's own text."""
    try:
        lib = load_library()
    except Exception:       # noqa: BLE001 -- host-only use without the library: Python formatting
        return 0
    return C.cast(lib.thr_format_toad, C.c_void_p).value or 0


class Ticket(object):
    """An open thr_submit*(): the records array the library will fill and the inputs it may
    still be reading (kept alive here)."""
    __slots__ = ("id", "out", "keep")

    def __init__(self, tid, out, keep):
        self.id, self.out, self.keep = tid, out, keep


FREQ_RANGE_DTYPE = np.dtype([("rxid", "<i4"), ("txid", "<i4"), ("lo", "<f8"), ("hi", "<f8")])


def identify(rxid, block, timestamp, carrier_bin, carrier_offset, energy, freq_ranges=None,
             device_id=0):
    """thr_identify on columns -> (txid int32[n], keep bool[n], kept_order int64[k]).
    freq_ranges: None (automatic windows) or rows (rxid, txid, lo, hi) in map order."""
    lib = load_library()
    n = len(rxid)
    cols = [np.ascontiguousarray(rxid, dtype=np.int32), np.ascontiguousarray(block, dtype=np.int32),
            np.ascontiguousarray(timestamp, dtype=np.float64),
            np.ascontiguousarray(carrier_bin, dtype=np.int32),
            np.ascontiguousarray(carrier_offset, dtype=np.float64),
            np.ascontiguousarray(energy, dtype=np.float64)]
    assert all(len(c) == n for c in cols)
    fmap = (np.zeros(0, dtype=FREQ_RANGE_DTYPE) if freq_ranges is None
            else np.ascontiguousarray(np.asarray(freq_ranges, dtype=FREQ_RANGE_DTYPE)))
    if freq_ranges is not None and len(fmap) == 0:
        raise ValueError("empty frequency map")
    txid = np.zeros(n, dtype=np.int32)
    keep = np.zeros(n, dtype=np.uint8)
    order = np.zeros(n, dtype=np.int64)
    n_kept = C.c_size_t(0)
    _check(lib, lib.thr_identify(int(device_id), n, *[c.ctypes.data for c in cols],
                                 fmap.ctypes.data if len(fmap) else None, len(fmap),
                                 txid.ctypes.data, keep.ctypes.data, order.ctypes.data,
                                 C.byref(n_kept)))
    return txid, keep.astype(bool), order[:n_kept.value]


class HostPin(object):
    """thr_host_register over a bytes-like object (an mmap of the input file): page-locks it so
    that the engine's chunk copies are asynchronous DMA out of the page cache.  Best effort:
    `.ok` False (and nothing locked) if the range is larger than `limit` bytes -- by default a
    quarter of the memory the kernel says is available -- or the runtime refuses; the engine
    works the same on unlocked memory.  close() (or garbage collection) unlocks."""

    def __init__(self, buf, limit=None):
        self._ptr, self.ok, self.why = None, False, ""
        arr = np.frombuffer(buf, dtype=np.uint8)
        if limit is None:
            limit = _available_memory() // 4
        if arr.size == 0 or arr.size > limit:
            self.why = "%d bytes against a limit of %d" % (arr.size, limit)
            return
        try:
            lib = load_library()
            rc = lib.thr_host_register(arr.ctypes.data, arr.size)
        except Exception as exc:       # no library / no device: the caller's engine will say so itself
            self.why = str(exc)
            return
        if rc != 0:
            self.why = lib.thr_last_error().decode()
            return
        self._lib, self._ptr, self.ok = lib, arr.ctypes.data, True

    def close(self):
        if self._ptr is not None:
            self._lib.thr_host_unregister(self._ptr)
            self._ptr, self.ok = None, False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _available_memory():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except (OSError, ValueError):
        pass
    return 1 << 33


def plan_sections(block_len, history_len, template_len):
    """thr_plan_sections: the overlap-save plan of a long block's correlate stage (host-only).
    Returns a list of dicts (start, win_lo, win_hi, sum_lo, sum_hi), block coordinates; [] when the
    geometry is not sectioned."""
    lib = load_library()
    n = C.c_int()
    arrs = [(C.c_int * 8)() for _ in range(5)]
    _check(lib, lib.thr_plan_sections(int(block_len), int(history_len), int(template_len), C.byref(n), *arrs))
    keys = ("start", "win_lo", "win_hi", "sum_lo", "sum_hi")
    return [dict((k, int(a[g])) for k, a in zip(keys, arrs)) for g in range(n.value)]


# Handles still open when the interpreter exits are destroyed by an atexit hook -- threads joined,
# pages unlocked, streams gone BEFORE the process runs the HIP runtime's own static destructors.  (A
# Detector's stage objects refer back to it, so a script that never calls close() keeps its engine,
# and its input window's threads inside the runtime, until then.)
_live_engines = weakref.WeakSet()


def _close_live_engines():
    for eng in list(_live_engines):
        try:
            eng.close()
        except Exception:       # noqa: BLE001 -- at exit: nothing to report to
            pass


atexit.register(_close_live_engines)


class Engine(object):
    """One detector handle == one (device, stream).  Not thread-safe per handle."""

    def __init__(self, block_len, history_len, templates, carrier_thresh, carrier_window,
                 corr_thresh, carrier_len=0, device_id=0, max_batch=256, preshift_num=0,
                 fastdet=False, path="auto", interpolator="parabolic"):
        """preshift_num > 0 selects the PreshiftDetector variant (thr_create_preshift);
        fastdet=True the fastdet-compatible one (thr_create_fastdet, power-domain thresholds).
        path: "auto" (the fastest kernels for the block length), "multipass" (the generic
        multi-pass pipeline whatever the length) or "unsectioned" (block_len 32768 / 65536: one
        long transform pair instead of overlap-save sections) -- thr_create_ex's THR_PATH_*; the
        non-default paths are independent implementations kept for cross-checks.
        interpolator (preshift variant): "parabolic" | "none" | "gaussian" | "cosine" (THR_INTERP_*)."""
        lib = load_library()
        tpl = np.ascontiguousarray(np.atleast_2d(np.asarray(templates, dtype=np.float64)))
        if tpl.ndim != 2:
            raise ValueError("templates must be 1-D or [n_templates, template_len]")
        st = ThrSettings()
        st.block_len, st.history_len = int(block_len), int(history_len)
        st.n_templates, st.template_len = tpl.shape
        st.templates = tpl.ctypes.data_as(C.POINTER(C.c_double))
        st.carrier_len = int(carrier_len)
        window = (0, -1) if carrier_window is None else carrier_window
        st.carrier_window[0], st.carrier_window[1] = int(window[0]), int(window[1])
        for i in range(3):
            st.carrier_thresh[i] = float(carrier_thresh[i])
            st.corr_thresh[i] = float(corr_thresh[i])
        st.device_id, st.max_batch = int(device_id), int(max_batch)
        handle = C.c_void_p()
        variant = VARIANT_FASTDET if fastdet else VARIANT_PRESHIFT if preshift_num else VARIANT_DEFAULT
        arg = int(preshift_num) | (INTERPOLATORS[interpolator] << 16 if preshift_num and not fastdet else 0)
        _check(lib, lib.thr_create_ex(C.byref(st), variant, arg, PATHS[path], C.byref(handle)))
        self.path = path
        self.preshift_num = int(preshift_num)
        self._lib, self._h = lib, handle
        _live_engines.add(self)
        self.block_len, self.n_templates = int(block_len), int(tpl.shape[0])
        self.history_len = int(history_len)
        self.max_batch = int(max_batch)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.thr_destroy(self._h)      # (closes an input window too)
            self._h = None
            self._window = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer path -------------------------------------------------
    def _as_input(self, blocks):
        a = np.asarray(blocks)
        if a.dtype == np.uint8:
            a = np.ascontiguousarray(a).reshape(-1, 2 * self.block_len)
            return a, THR_IN_U8
        a = np.ascontiguousarray(a.astype(np.complex64, copy=False)).reshape(-1, self.block_len)
        return a, THR_IN_C64

    def detect(self, blocks, block_idx=None):
        """blocks: u8 [B, 2N] or complex64 [B, N] -> structured records [B, n_templates]."""
        a, fmt = self._as_input(blocks)
        nb = a.shape[0]
        out = np.zeros((nb, self.n_templates), dtype=RECORD_DTYPE)
        idx_p = None
        if block_idx is not None:
            idx = np.ascontiguousarray(np.asarray(block_idx, dtype=np.int64))
            assert idx.shape == (nb,)
            idx_p = idx.ctypes.data
        _check(self._lib, self._lib.thr_detect(self._h, a.ctypes.data, fmt, idx_p, nb,
                                               out.ctypes.data))
        return out

    def detect_offsets(self, blocks, carrier_offset, block_idx=None):
        """thr_detect_offsets: detect() with the sub-bin carrier offset of every block given (float64
        [B]) instead of fitted -- the slow path behind a replaced `Detector.sync.interpolator`."""
        a, fmt = self._as_input(blocks)
        nb = a.shape[0]
        off = np.ascontiguousarray(np.asarray(carrier_offset, dtype=np.float64))
        assert off.shape == (nb,)
        out = np.zeros((nb, self.n_templates), dtype=RECORD_DTYPE)
        idx, idx_p = self._idx_ptr(block_idx, nb)
        _check(self._lib, self._lib.thr_detect_offsets(self._h, a.ctypes.data, fmt, idx_p, nb, off.ctypes.data,
                                                       out.ctypes.data))
        return out

    def detect_card(self, text, payload_off, block_idx=None):
        """text: bytes-like holding .card records; payload_off: int64 offsets of each block's
        base64 payload.  Decoding happens on the device.  -> records [B, n_templates]."""
        buf = np.frombuffer(text, dtype=np.uint8)
        off = np.ascontiguousarray(np.asarray(payload_off, dtype=np.int64))
        nb = off.shape[0]
        out = self._records_out(nb)
        idx_p = None
        if block_idx is not None:
            idx = np.ascontiguousarray(np.asarray(block_idx, dtype=np.int64))
            assert idx.shape == (nb,)
            idx_p = idx.ctypes.data
        _check(self._lib, self._lib.thr_detect_card(self._h, buf.ctypes.data, buf.size,
                                                    off.ctypes.data, idx_p, nb, out.ctypes.data))
        return out

    def detect_stream(self, stream, first_block_idx=0):
        """stream: bytes-like raw interleaved u8 I/Q; the overlapping blocks
        (block_reader framing, stride block_len - history_len samples) are framed on the
        device.  -> records [n_whole_blocks, n_templates]."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        stride = 2 * (self.block_len - self.history_len)
        nb = 0 if buf.size < 2 * self.block_len else (buf.size - 2 * self.block_len) // stride + 1
        out = self._records_out(nb)
        got = C.c_size_t(0)
        _check(self._lib, self._lib.thr_detect_stream(self._h, buf.ctypes.data, buf.size,
                                                      int(first_block_idx), out.ctypes.data, nb,
                                                      C.byref(got)))
        assert got.value == nb
        return out

    # ---- asynchronous host path: submit a batch, collect its records later --
    # Record arrays of the ticket interface.  A batch's array is 128 KiB and up -- glibc serves that by
    # mmap, and every mmap / munmap / first-touch fault of a process whose input window is page-locking
    # a file waits for the address-space lock the lockers hold for milliseconds at a time (the iteration
    # over a .card file ran anywhere between 0.66 and 1.05 M blocks/s).  A caller that is done with a
    # collected array hands it back (`recycle`) and the next submit reuses its pages.
    def _records_out(self, nb):
        pool = self.__dict__.setdefault("_out_pool", {})
        spare = pool.get(nb)
        if spare:
            return spare.pop()
        return np.zeros((nb, self.n_templates), dtype=RECORD_DTYPE)

    def recycle(self, records):
        """Hand a collected record array back for reuse by a later submit*() (the caller keeps no view of it)."""
        base = records
        while isinstance(getattr(base, "base", None), np.ndarray):
            base = base.base
        if (isinstance(base, np.ndarray) and base.dtype == RECORD_DTYPE and base.ndim == 2
                and base.shape[1] == self.n_templates and base.flags.c_contiguous and base.flags.owndata):
            spare = self.__dict__.setdefault("_out_pool", {}).setdefault(base.shape[0], [])
            if len(spare) < MAX_IN_FLIGHT + 2:
                spare.append(base)

    def _idx_ptr(self, block_idx, nb):
        if block_idx is None:
            return None, None
        idx = np.ascontiguousarray(np.asarray(block_idx, dtype=np.int64))
        assert idx.shape == (nb,)
        return idx, idx.ctypes.data

    def submit(self, blocks, block_idx=None):
        """thr_submit: like detect() for ONE batch (<= max_batch blocks), but returns a Ticket at
        once; `collect(ticket)` -> records [B, n_templates].  Up to MAX_IN_FLIGHT may be open."""
        a, fmt = self._as_input(blocks)
        nb = a.shape[0]
        out = self._records_out(nb)
        idx, idx_p = self._idx_ptr(block_idx, nb)
        t = C.c_uint64(0)
        _check(self._lib, self._lib.thr_submit(self._h, a.ctypes.data, fmt, idx_p, nb, out.ctypes.data,
                                               C.byref(t)))
        return Ticket(t.value, out, (a, idx))

    def submit_card(self, text, payload_off, block_idx=None):
        """thr_submit_card (see detect_card)."""
        buf = np.frombuffer(text, dtype=np.uint8)
        off = np.ascontiguousarray(np.asarray(payload_off, dtype=np.int64))
        nb = off.shape[0]
        out = self._records_out(nb)
        idx, idx_p = self._idx_ptr(block_idx, nb)
        t = C.c_uint64(0)
        _check(self._lib, self._lib.thr_submit_card(self._h, buf.ctypes.data, buf.size, off.ctypes.data,
                                                    idx_p, nb, out.ctypes.data, C.byref(t)))
        # (`text` itself is NOT kept: a reader's bytearray must stay resizable -- the caller keeps
        # it valid until inputs_consumed() / collect())
        return Ticket(t.value, out, (off, idx))

    def submit_stream(self, stream, first_block_idx=0):
        """thr_submit_stream (see detect_stream)."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        stride = 2 * (self.block_len - self.history_len)
        nb = 0 if buf.size < 2 * self.block_len else (buf.size - 2 * self.block_len) // stride + 1
        out = self._records_out(nb)
        got, t = C.c_size_t(0), C.c_uint64(0)
        _check(self._lib, self._lib.thr_submit_stream(self._h, buf.ctypes.data, buf.size,
                                                      int(first_block_idx), out.ctypes.data, nb,
                                                      C.byref(got), C.byref(t)))
        assert got.value == nb
        return Ticket(t.value, out, None)

    def input_window(self, buf=None, populate_threads=0, segment_bytes=0):
        """thr_input_window[_ex]: declare `buf` (bytes-like: the mmap of the input file, or a slice of
        it) as the range the host entry points will read front to back -- a library thread
        page-locks it a bounded distance ahead of the copies, which then are asynchronous DMA.
        None closes the window.  The caller keeps `buf` alive until then.  populate_threads: the
        page-table populators ahead of the locking (0 = the library's default, 3; a rank of a sharded
        run passes its share of the host, parallel.populate_threads); segment_bytes: the locking
        granularity (0 = 128 MiB; tests)."""
        if buf is None:
            _check(self._lib, self._lib.thr_input_window(self._h, None, 0))
            self._window = None
            return
        if getattr(self, "_window", None) is not None:     # (waits for a released window's unlocking)
            _check(self._lib, self._lib.thr_input_window(self._h, None, 0))
            self._window = None
        arr = np.frombuffer(buf, dtype=np.uint8)
        _check(self._lib, self._lib.thr_input_window_ex(self._h, arr.ctypes.data, arr.size,
                                                        int(populate_threads), int(segment_bytes)))
        self._window = arr

    def input_window_release(self):
        """thr_input_window_release: the input has been read -- stop locking, unlock what is still
        locked in the background, return at once.  This object keeps the buffer alive until the
        window is really closed (input_window(None), a new window, or close())."""
        self._lib.thr_input_window_release.argtypes = [C.c_void_p]
        _check(self._lib, self._lib.thr_input_window_release(self._h))

    # ---- the whole file -> .toad loop inside the library (thr_run_card / thr_run_stream) -------
    def _run_opts(self, out_fd, rxid, with_txid, carrier_offset_mode, batch_blocks, rec_out, timestamp=None):
        o = ThrRunOpts()
        o.struct_bytes = C.sizeof(ThrRunOpts)
        o.batch_blocks = int(batch_blocks or 0)
        o.out_fd = -1 if out_fd is None else int(out_fd)
        o.with_rxid, o.rxid = (0, 0) if rxid is None else (1, int(rxid))
        o.with_txid = int(bool(with_txid))
        o.carrier_offset_mode = int(carrier_offset_mode)
        o.timestamp = float("nan") if timestamp is None else float(timestamp)
        if rec_out is not None:
            assert rec_out.dtype == RECORD_DTYPE and rec_out.flags.c_contiguous
            o.rec_out, o.rec_capacity = rec_out.ctypes.data, rec_out.size
        return o

    def _run_done(self, rc, stats):
        """-> stats dict; THR_ERR_INDEX is reported in it (`index_error`), everything else raises."""
        out = stats.as_dict()
        out["index_error"] = rc == ERR_INDEX
        if rc != 0 and rc != ERR_INDEX:
            raise NativeError("libthriftyhip: %s (code %d)" % (self._lib.thr_last_error().decode(), rc),
                              rc, out)
        return out

    def run_card(self, text, out_fd=None, rxid=None, with_txid=False, carrier_offset_mode=0,
                 batch_blocks=0, rec_out=None):
        """thr_run_card: frame, detect and format the .card text `text` (bytes-like: the mapped
        file) inside the library; the .toad text goes to the descriptor `out_fd`, the detected
        records (timestamp bits in `reserved`) into `rec_out` (a RECORD_DTYPE array) if given.
        -> stats dict (blocks, detections, seconds per stage, `index_error` ...)."""
        buf = np.frombuffer(text, dtype=np.uint8)
        o = self._run_opts(out_fd, rxid, with_txid, carrier_offset_mode, batch_blocks, rec_out)
        st = ThrRunStats()
        rc = self._lib.thr_run_card(self._h, buf.ctypes.data, buf.size, C.byref(o), C.byref(st))
        del buf
        return self._run_done(rc, st)

    def run_stream(self, stream, first_block_idx=0, out_fd=None, rxid=None, with_txid=False,
                   carrier_offset_mode=0, batch_blocks=0, rec_out=None, timestamp=None):
        """thr_run_stream: the same for a raw u8 I/Q stream whose first 2 * block_len bytes are
        block `first_block_idx` (overlap framing on the device)."""
        buf = np.frombuffer(stream, dtype=np.uint8)
        o = self._run_opts(out_fd, rxid, with_txid, carrier_offset_mode, batch_blocks, rec_out, timestamp)
        st = ThrRunStats()
        rc = self._lib.thr_run_stream(self._h, buf.ctypes.data, buf.size, int(first_block_idx), C.byref(o),
                                      C.byref(st))
        del buf
        return self._run_done(rc, st)

    def debug_window(self):
        """thr_debug_window -> (released_below, locked_lo, locked_hi, segment_bytes), byte offsets
        from the window's page-aligned start (test hook)."""
        out = (C.c_size_t * 4)()
        self._lib.thr_debug_window.argtypes = [C.c_void_p, C.c_size_t * 4]
        _check(self._lib, self._lib.thr_debug_window(self._h, out))
        return tuple(int(v) for v in out)

    def correlate_geom(self):
        """thr_debug_correlate_geom -> (rows_lo, rows_hi) of the window-row specialisation this
        handle's correlate launches take, or (-1, -1): the generic kernel."""
        lo, hi = C.c_int(-1), C.c_int(-1)
        self._lib.thr_debug_correlate_geom.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _check(self._lib, self._lib.thr_debug_correlate_geom(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def path_info(self):
        """thr_get_path_info -> dict: which kernels this handle's plain launches take and why
        (`why_unsectioned` is one of WHY_UNSECTIONED; `text` says the same as one sentence)."""
        info = ThrPathInfo()
        self._lib.thr_get_path_info.argtypes = [C.c_void_p, C.POINTER(ThrPathInfo)]
        _check(self._lib, self._lib.thr_get_path_info(self._h, C.byref(info)))
        return {"n_sections": info.n_sections, "section_len": info.section_len,
                "rows": (info.rows_lo, info.rows_hi), "why_unsectioned": WHY_UNSECTIONED[info.why_unsectioned],
                "n_templates": info.n_templates, "carrier_kernel": info.carrier_kernel.decode(),
                "correlate_kernel": info.correlate_kernel.decode(), "text": info.text.decode()}

    def sections(self):
        """thr_debug_sections -> (n_sections, section_len) of this handle's plain correlate launches
        ((0, 0): unsectioned)."""
        n, ln = C.c_int(0), C.c_int(0)
        self._lib.thr_debug_sections.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _check(self._lib, self._lib.thr_debug_sections(self._h, C.byref(n), C.byref(ln)))
        return n.value, ln.value

    def debug_pipe_times(self):
        """thr_debug_pipe_times -> seconds per phase of the host entry points' chunks (reads and resets)."""
        out = (C.c_double * 16)()
        self._lib.thr_debug_pipe_times.argtypes = [C.c_void_p, C.c_double * 16]
        _check(self._lib, self._lib.thr_debug_pipe_times(self._h, out))
        keys = ("grow_s", "h2d_s", "meta_s", "launch_s", "d2h_s", "chunks", "event_s", "fill_s")
        res = dict(zip(keys, (float(v) for v in out[:8])))
        res.update(("max_" + k, float(v)) for k, v in zip(keys, out[8:]) if k != "chunks")
        return res

    def debug_window_times(self):
        """thr_debug_window_times -> dict of seconds per activity of the input window's threads."""
        out = (C.c_double * 6)()
        self._lib.thr_debug_window_times.argtypes = [C.c_void_p, C.c_double * 6]
        _check(self._lib, self._lib.thr_debug_window_times(self._h, out))
        keys = ("populate_s", "register_s", "unregister_s", "acquire_wait_s", "acquire_waits", "pageable_copies")
        return dict(zip(keys, (float(v) for v in out)))

    def set_wait_mode(self, sleeping):
        """thr_set_wait_mode: True = collect() naps between queries of the batch's event instead of
        polling it (a CPU per rank saved on hosts shared by several ranks)."""
        self._lib.thr_set_wait_mode.argtypes = [C.c_void_p, C.c_int]
        _check(self._lib, self._lib.thr_set_wait_mode(self._h, int(bool(sleeping))))

    def collect(self, ticket):
        """thr_collect: wait for the ticket's batch -> its records [B, n_templates]."""
        _check(self._lib, self._lib.thr_collect(self._h, ticket.id))
        ticket.keep = None
        return ticket.out

    def inputs_consumed(self, ticket):
        """thr_inputs_consumed: wait until the ticket's input arrays may be overwritten."""
        _check(self._lib, self._lib.thr_inputs_consumed(self._h, ticket.id))
        ticket.keep = None

    def poll(self, ticket):
        done = C.c_int(0)
        _check(self._lib, self._lib.thr_poll(self._h, ticket.id, C.byref(done)))
        return bool(done.value)

    # ---- device-resident path (pointers are plain integers) ---------------
    def detect_stream_device(self, d_stream, n_blocks, d_out, d_block_idx=None):
        _check(self._lib, self._lib.thr_detect_stream_device(self._h, d_stream, d_block_idx,
                                                             n_blocks, d_out))

    def detect_device(self, d_samples, fmt, n_blocks, d_out, d_block_idx=None):
        _check(self._lib, self._lib.thr_detect_device(self._h, d_samples, fmt, d_block_idx,
                                                      n_blocks, d_out))

    def compact_device(self, d_in, n_records, d_out):
        kept = C.c_size_t(0)
        _check(self._lib, self._lib.thr_compact_device(self._h, d_in, n_records, d_out,
                                                       C.byref(kept)))
        return kept.value

    def sync(self):
        _check(self._lib, self._lib.thr_sync(self._h))

    def set_stream(self, stream_ptr):
        """Run on the caller's HIP stream.  0 / None -- the handle value of torch's DEFAULT stream
        -- selects the device's legacy default stream (ordered with torch's fills and copies
        there); `use_own_stream()` goes back to the engine's private non-blocking stream."""
        if not stream_ptr:
            _check(self._lib, self._lib.thr_set_stream_default(self._h))
        else:
            _check(self._lib, self._lib.thr_set_stream(self._h, stream_ptr))

    def use_own_stream(self):
        _check(self._lib, self._lib.thr_set_stream(self._h, None))

    def profile_enable(self, every=1):
        """every = n > 0: time the kernels of every n-th batch; 0/False: off."""
        _check(self._lib, self._lib.thr_profile_enable(self._h, int(every)))

    def profile_read(self):
        ms = (C.c_double * N_KERNEL_SLOTS)()
        cnt = (C.c_int64 * N_KERNEL_SLOTS)()
        _check(self._lib, self._lib.thr_profile_read(self._h, ms, cnt))
        return {self._lib.thr_kernel_name(i).decode(): (ms[i], cnt[i])
                for i in range(N_KERNEL_SLOTS)}

    # ---- test hooks ------------------------------------------------------
    def debug_fft(self, blocks):
        a, fmt = self._as_input(blocks)
        out = np.zeros((a.shape[0], self.block_len), dtype=np.complex64)
        _check(self._lib, self._lib.thr_debug_fft(self._h, a.ctypes.data, fmt, a.shape[0],
                                                  out.ctypes.data))
        return out

    def debug_stage(self, blocks, template_id=0, carrier_offset=None):
        a, fmt = self._as_input(blocks)
        xhat = np.zeros((a.shape[0], self.block_len), dtype=np.complex64)
        corr = np.zeros((a.shape[0], self.block_len), dtype=np.complex64)
        off = None
        if carrier_offset is not None:
            off = np.ascontiguousarray(carrier_offset, dtype=np.float64)
            if off.shape != (a.shape[0],):
                raise ValueError("carrier_offset: one value per block")
        _check(self._lib, self._lib.thr_debug_stage_offsets(self._h, a.ctypes.data, fmt, a.shape[0],
                                                            template_id, off.ctypes.data if off is not None else None,
                                                            xhat.ctypes.data, corr.ctypes.data))
        return xhat, corr
