"""Build the gfx950 shared library in-tree (no torch involved).

    python -m thrifty_amd.build            # hipcc -> thrifty_amd/libthriftyhip.so
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libthriftyhip.so")
SOURCES = ["handle.hip", "window.hip", "pipeline.hip", "entry.hip", "text.hip", "detect16k.hip", "detect16k_geom0.hip", "detect16k_geom1.hip", "detect16k_geom2.hip", "detect16k_carrier.hip", "detect16k_preshift.hip", "detect16k_sec.hip", "detect_seg.hip", "detect_long.hip", "detect_small.hip", "generic.hip", "card_ingest.hip", "identify.hip", "run_file.hip"]
HEADERS = ["host_internal.hpp", "correlate16k.hpp", "correlate16k_geom.hpp", "detect_common.hpp", "fft_regs.hpp", "kernel_util.hpp", "lmdif8.hpp", "passes_w8.hpp", os.path.join("..", "..", "include", "thrifty_hip.h")]
HOST_ONLY = ("handle.hip", "window.hip", "pipeline.hip", "entry.hip", "text.hip", "run_file.hip",
             "host_internal.hpp")     # no kernels: not part of csrc_hash()
# per-file code-generation flags (measured on MI355X, see csrc/detect16k_carrier.hip)
PER_FILE_FLAGS = {"detect16k_carrier.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
                  # the work cursor's atomicAdd stays ONE lane's atomic whose result is waited for where it is
                  # used (the wave-aggregation pass puts a v_readfirstlane, and with it the wait, right behind it)
                  "detect16k_sec.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


# host glue in C (CPython API, no device code): the batch constructor of the per-block result objects
FAST_SRC = os.path.join(CSRC, "fastresults.c")
FAST_LIB = os.path.join(HERE, "_fastresults.so")


def build_fastresults(force=False, verbose=False):
    """gcc -> thrifty_amd/_fastresults.so (imported as thrifty_amd._fastresults)."""
    import sysconfig
    if not force and os.path.exists(FAST_LIB) and os.path.getmtime(FAST_LIB) > max(
            os.path.getmtime(FAST_SRC), os.path.getmtime(os.path.abspath(__file__))):
        return FAST_LIB
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler: thrifty_amd._fastresults cannot be built")
    cmd = [cc, "-O2", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"], FAST_SRC, "-o", FAST_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return FAST_LIB


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the MI355X engine cannot be built")
    return exe


def csrc_hash():
    """sha256 (first 16 hex digits) over the kernel sources and headers (everything but the host
    side, HOST_ONLY): profiles/hbm_traffic.json records the hash its counter passes were taken on,
    bench.py flags a mismatch (`traffic_stale`)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted([x for x in SOURCES if x not in HOST_ONLY] + [x for x in HEADERS if not x.startswith("..")]):
        h.update(name.encode())
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    try:       # objects / library of another flag set (a dev variant): rebuild
        if open(os.path.join(CSRC, ".build_flags")).read() != " ".join(os.environ.get("THR_EXTRA_CFLAGS", "").split()):
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libthriftyhip.so (and the C result builder)."""
    build_fastresults(force=force, verbose=verbose)
    if not force and not needs_build():
        return LIB
    # -fno-slp-vectorize: the kernels are hand-vectorised with ext-vector float2;
    # the SLP vectorizer only adds v_mov shuffles (see csrc/fft_regs.hpp)
    extra = os.environ.get("THR_EXTRA_CFLAGS", "").split()
    jobs = []
    objs = []
    # an object is rebuilt when its source, any shared header or this file is newer than it (every
    # translation unit includes the shared headers) -- or when the objects on disk were compiled
    # with a different flag set: the stamp file names the THR_EXTRA_CFLAGS they belong to, so a
    # default build after a dev-variant build (scripts/build_variant.sh) never links leftovers
    common = [os.path.join(CSRC, hname) for hname in HEADERS] + [os.path.abspath(__file__)]
    newest_common = max(os.path.getmtime(d) for d in common)
    stamp = os.path.join(CSRC, ".build_flags")
    flags_now = " ".join(extra)
    try:
        flags_then = open(stamp).read()
    except OSError:
        flags_then = None
    if flags_then != flags_now:
        # objects of another flag set: none of them may survive, whatever interrupts this rebuild --
        # the library and every object go first, the stamp is written LAST (after the link), so an
        # interrupted rebuild is still seen as "flags differ" by the next one
        force = True
        for stale in [LIB, stamp] + [os.path.join(CSRC, src.replace(".hip", ".o")) for src in SOURCES]:
            if os.path.exists(stale):
                os.remove(stale)
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj) and
                os.path.getmtime(obj) > max(newest_common, os.path.getmtime(os.path.join(CSRC, src)))):
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-fno-slp-vectorize"] + PER_FILE_FLAGS.get(src, []) + extra + [
                   "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((subprocess.Popen(cmd), cmd, obj))   # the translation units build in parallel
    for proc, cmd, obj in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
