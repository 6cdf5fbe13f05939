"""Block framing: raw RTL-SDR byte streams and .card text -> IQ blocks.

Python-3 counterpart of reference thrifty/block_data.py.  The readers yield the
same `(timestamp, block_idx, samples)` tuples; `samples` is an :class:`IQBlock`
-- a complex ndarray that also remembers the interleaved u8 bytes it came from,
so the GPU engine can ingest 2 bytes/sample instead of 8 (the conversion
`(u8 - 127.4) / 128` then happens inside the first FFT pass on the device).
"""
from __future__ import annotations

import base64
import time

import numpy as np

_SKIP_PREFIXES = ("Using Volk machine:", "linux;")


class IQBlock(np.ndarray):
    """complex ndarray view with an optional `.raw` (uint8 I,Q interleaved) twin."""

    def __new__(cls, samples, raw=None):
        obj = np.asarray(samples).view(cls)
        obj.raw = raw
        return obj

    def __array_finalize__(self, obj):
        # derived arrays (slices, arithmetic results) no longer match the raw bytes
        self.raw = None


def raw_to_complex(data):
    """u8 I/Q pairs -> complex64, (v - 127.4) / 128 (reference block_data.py:38-52)."""
    values = np.asarray(data, dtype=np.uint8).astype(np.float32).view(np.complex64)
    values -= 127.4 + 127.4j
    values /= 128
    return values


def complex_to_raw(array):
    """Inverse quantiser: *128 + 127.4, truncate to u8 (reference block_data.py:55-67)."""
    scaled = np.asarray(array).astype(np.complex64).view(np.float32) * 128 + 127.4
    return scaled.astype(np.uint8)


def _fixed_chunks(stream, nbytes):
    """Yield exactly-`nbytes` chunks; a short tail is dropped (as the reference does)."""
    pending = b""
    while True:
        buf = stream.read(nbytes - len(pending))
        if not buf:
            return
        pending += buf
        if len(pending) < nbytes:
            continue
        yield np.frombuffer(pending, dtype=np.uint8)
        pending = b""


def block_reader(stream, size, history):
    """Overlapping blocks from a raw u8 I/Q stream (reference block_data.py:70-98).

    Each block holds `history` samples of the previous block followed by
    `size - history` new ones; the very first history is 0.0 (not quantiser
    zero), exactly like the reference, so block 0 has no u8 twin.
    """
    new = size - history
    data = np.zeros(size)
    raw_hist = None
    for block_idx, chunk in enumerate(_fixed_chunks(stream, new * 2)):
        data = np.concatenate([data[-history:] if history else data[:0], raw_to_complex(chunk)])
        raw = None
        if raw_hist is not None or history == 0:
            raw = chunk.copy() if history == 0 else np.concatenate([raw_hist, chunk])
        yield time.time(), block_idx, IQBlock(data, raw)
        if history:
            tail = chunk[-2 * history:] if 2 * history <= len(chunk) else None
            if tail is not None:
                raw_hist = tail
            elif raw is not None:
                raw_hist = raw[-2 * history:]
            else:
                raw_hist = None


def card_reader(stream):
    """Blocks from a .card stream: `<timestamp> <block_idx> <base64 u8 I/Q>` per line;
    `#` comments, blank lines and fastcard's banner lines are skipped
    (reference block_data.py:101-131)."""
    while True:
        line = stream.readline()
        if len(line) == 0:
            return
        if isinstance(line, bytes):
            line = line.decode("ascii")
        if line[0] in "#\n\r":
            continue
        if line.startswith(_SKIP_PREFIXES):
            continue
        timestamp, idx, encoded = line.rstrip("\r\n").split(" ")
        raw = np.frombuffer(base64.b64decode(encoded), dtype=np.uint8)
        yield float(timestamp), int(idx), IQBlock(raw_to_complex(raw), raw)


def card_line(timestamp, block_idx, raw):
    """Format one .card line (fastcard_cli.c:187-192: "%ld.%06ld %PRId64 %s\\n")."""
    sec = int(timestamp)
    usec = int(round((timestamp - sec) * 1e6))
    return "%d.%06d %d %s\n" % (sec, usec, block_idx,
                                base64.b64encode(np.asarray(raw, np.uint8).tobytes()).decode())
