"""Block framing: raw RTL-SDR byte streams and .card text -> IQ blocks.

Python-3 counterpart of reference thrifty/block_data.py.  The readers yield the
same `(timestamp, block_idx, samples)` tuples; `samples` is an :class:`IQBlock`
-- a complex ndarray that also remembers the interleaved u8 bytes it came from,
so the GPU engine can ingest 2 bytes/sample instead of 8 (the conversion
`(u8 - 127.4) / 128` then happens inside the first FFT pass on the device).
"""
from __future__ import annotations

import base64
import mmap
import os
import select
import stat
import time

import numpy as np

_SKIP_PREFIXES = ("Using Volk machine:", "linux;")
_frame_card = None      # thr_frame_card binding (resolved on first use; False: library not available)


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
    """Yield exactly-`nbytes` chunks; a short tail is dropped (as the reference does).  One
    preallocated buffer per chunk, filled in place (a pipe delivers a chunk in pieces)."""
    fill = getattr(stream, "readinto", None)
    while True:
        chunk = np.empty(nbytes, dtype=np.uint8)
        have = 0
        while have < nbytes:
            if fill is not None:
                got = fill(memoryview(chunk)[have:]) or 0
            else:
                piece = stream.read(nbytes - have)
                got = len(piece)
                chunk[have:have + got] = np.frombuffer(piece, dtype=np.uint8)
            if got == 0:
                return
            have += got
        yield chunk


def is_live(stream):
    """True if `stream` reads from a pipe, socket or terminal -- a source that delivers blocks as
    they are captured (a regular file or an in-memory stream is not)."""
    try:
        mode = os.fstat(stream.fileno()).st_mode
    except (AttributeError, OSError, ValueError):
        return False
    return stat.S_ISFIFO(mode) or stat.S_ISSOCK(mode) or stat.S_ISCHR(mode)


class _BlockIterator(object):
    """The classic readers' generator plus `.live` (is_live of the stream): a batching consumer
    stops filling a batch when a live source makes it wait (thrifty_amd.detect.Detector)."""

    def __init__(self, gen, live):
        self._gen, self.live = gen, live

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)

    next = __next__


def block_reader(stream, size, history):
    """Overlapping blocks from a raw u8 I/Q stream (reference block_data.py:70-98).

    Each block holds `history` samples of the previous block followed by
    `size - history` new ones; the very first history is 0.0 (not quantiser
    zero), exactly like the reference, so block 0 has no u8 twin.
    """
    return _BlockIterator(_block_reader(stream, size, history), is_live(stream))


def _block_reader(stream, size, history):
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
    return _BlockIterator(_card_reader(stream), is_live(stream))


def _is_data_line(text):
    """Not a comment (`#`), a blank line or one of the capture tool's banner lines."""
    return text[0] not in "#\n\r" and not text.startswith(_SKIP_PREFIXES)


def _lines(stream):
    """Lines of `stream` as str until readline() returns nothing (text or binary streams)."""
    for ln in iter(stream.readline, None):
        if not ln:
            return
        yield ln.decode("ascii") if isinstance(ln, bytes) else ln


def _card_reader(stream):
    for text in filter(_is_data_line, _lines(stream)):
        header = text.rstrip("\r\n").split(" ")
        if len(header) != 3:
            raise ValueError("not enough values to unpack (expected 3, got %d)" % len(header)
                             if len(header) < 3 else "too many values to unpack (expected 3)")
        raw = np.frombuffer(base64.b64decode(header[2]), dtype=np.uint8)
        yield float(header[0]), int(header[1]), IQBlock(raw_to_complex(raw), raw)


def _map_regular_file(stream):
    """mmap of `stream` if it is a regular, seekable, non-empty binary file (else None): the
    batch readers then frame records straight out of the page cache and the H2D copy is the
    only pass over the bytes."""
    try:
        fd = stream.fileno()
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_size == 0 or "b" not in getattr(stream, "mode", "b"):
            return None, 0
        return mmap.mmap(fd, 0, access=mmap.ACCESS_READ), stream.tell()
    except (AttributeError, OSError, ValueError):
        return None, 0


def _readable_within(stream, seconds):
    """True if `stream`'s descriptor has data (or EOF) pending within `seconds`; streams
    without a descriptor (BytesIO, wrappers) read as 'nothing more pending'."""
    try:
        return bool(select.select([stream.fileno()], [], [], seconds)[0])
    except (AttributeError, OSError, ValueError):
        return False


def _read_arrived(stream, view, grace=0.002):
    """Fill `view` with what HAS ARRIVED: block for the first bytes (or EOF), then keep taking
    whatever shows up within `grace` seconds of the previous read.  -> bytes read (0 = EOF).

    `readinto` on a BufferedReader (sys.stdin.buffer) blocks until the whole view is full, so
    `fastcard ... | thrifty detect -` would emit nothing until 64 MiB had accumulated; the
    reference's readers return per line / per block.  A fast producer (`cat rx.card |`) keeps the
    descriptor readable, so its batches still fill up.  In-memory streams (no descriptor) have
    nothing to wait for: one plain `readinto`."""
    one = getattr(stream, "readinto1", None)
    try:
        stream.fileno()
    except (AttributeError, OSError, ValueError):
        one = None
    if one is None:
        if hasattr(stream, "readinto"):
            return stream.readinto(view) or 0
        more = stream.read(len(view))
        if isinstance(more, str):
            more = more.encode("ascii")
        view[:len(more)] = more
        return len(more)
    got = one(view) or 0
    total = got
    while got and total < len(view) and _readable_within(stream, grace):
        got = one(view[total:]) or 0
        total += got
    return total


def _widen_pipe(stream, size=1 << 20):
    """A FIFO's kernel buffer is 64 KiB by default -- one .card line is 43 KiB: ask for 1 MiB so
    that a fast producer is drained in few system calls (Linux F_SETPIPE_SZ; best effort)."""
    try:
        fd = stream.fileno()
        if stat.S_ISFIFO(os.fstat(fd).st_mode):
            import fcntl
            fcntl.fcntl(fd, getattr(fcntl, "F_SETPIPE_SZ", 1031), size)
    except (AttributeError, OSError, ValueError, ImportError):
        pass


class CardStream(object):
    """Batch-oriented .card reader for the GPU engine.

    The host only *finds* the records -- line ends, the two header fields, the offset of
    the base64 payload -- in large binary chunks; the payload text itself is handed to
    the device untouched (`Engine.detect_card`) and decoded there.  Skips the same lines
    as `card_reader` (comments, blanks, fastcard banners).  Also iterable as a plain
    `(timestamp, block_idx, IQBlock)` reader (host decode) for drop-in use.
    """

    def __init__(self, stream, block_len, chunk_bytes=64 << 20):
        self.stream = stream
        self.block_len = int(block_len)
        self.payload_chars = ((2 * self.block_len + 2) // 3) * 4
        line_max = self.payload_chars + 128
        self.chunk_bytes = max(int(chunk_bytes), 2 * line_max)
        # ONE reusable buffer: refills move the unconsumed tail (< one line) to the front and
        # read the next chunk in place -- no per-chunk reallocation / concatenation copies
        mapped, at = _map_regular_file(stream)
        if mapped is not None:       # regular file: the whole text is "already read"
            self._buf, self._pos, self._end, self._eof = mapped, at, len(mapped), True
            return
        _widen_pipe(stream)
        self._buf = bytearray(self.chunk_bytes)
        self._pos = 0   # first unconsumed byte
        self._end = 0   # one past the last valid byte
        self._eof = False

    @property
    def mapped(self):
        """True if the input is a regular file read through mmap (no read buffer of our own)."""
        return isinstance(self._buf, mmap.mmap)

    def _has_complete_record(self):
        """A whole DATA line (newline included) is buffered at or after the read position."""
        p, end, buf = self._pos, self._end, self._buf
        while p < end:
            k = buf.find(b"\n", p, end)
            if k < 0:
                return False
            c = buf[p]
            if not (c in (0x23, 0x0A, 0x0D) or buf[p:p + 19] == b"Using Volk machine:" or buf[p:p + 6] == b"linux;"):
                return True
            p = k + 1       # comment / blank / banner line: next_batch skips it
        return False

    def mapped_span(self):
        """The bytes of the mapped file this reader will hand out (its shard), as a memoryview --
        what a consumer may page-lock for DMA (thrifty_amd._native.HostPin); None if not mapped."""
        if not self.mapped:
            return None
        return memoryview(self._buf)[self._pos:self._end]

    def take_rest(self):
        """Mapped file only: everything this reader has not handed out yet (its shard), as ONE
        memoryview of .card text -- for a consumer that frames it itself (thr_run_card); the reader
        is exhausted afterwards.  -> (view, upper bound on the number of records in it)."""
        if not self.mapped:
            raise ValueError("take_rest() needs a mapped regular file")
        view = memoryview(self._buf)[self._pos:self._end]
        self._pos = self._end
        return view, len(view) // (self.payload_chars + 4) + 1

    def ready(self, release=None):
        """True if next_batch() would not have to WAIT for the source: a mapped file, the end of
        the stream, or a COMPLETE record in the buffer -- a partial line (a producer that does not
        flush at line ends) is not one.  Bytes waiting in the pipe are taken in first (what has
        arrived, never blocking); that rewrites the buffer the previous batch's text lives in, so
        `release` -- a callable that returns once the engine has copied that text -- is called
        before.  A Detector only reads ahead of the batch it is about to hand out when this holds:
        on a live pipe the results of what HAS arrived must not wait for the next line."""
        if self.mapped or self._eof or self._has_complete_record():
            return True
        if not _readable_within(self.stream, 0):
            return False
        if release is not None:
            release()
        self._fill()
        return self._eof or self._has_complete_record()

    def shard(self, rank, world):
        """Restrict a mapped .card file to the rank-th of `world` contiguous byte ranges, cut at
        line starts (a line belongs to the range its first byte lies in).  Lines have one length,
        so the ranges hold equal block counts to within one, and concatenating the ranks'
        outputs in rank order reproduces the file's order."""
        if world <= 1:
            return self
        if not isinstance(self._buf, mmap.mmap):
            raise ValueError("only a regular file can be sharded over GPUs (not a pipe)")
        start, end = self._pos, self._end

        def line_start(p):
            if p <= start:
                return start
            k = self._buf.find(b"\n", p - 1, end)
            return end if k < 0 else k + 1

        span = end - start
        self._pos = line_start(start + span * rank // world)
        self._end = line_start(start + span * (rank + 1) // world) if rank + 1 < world else end
        return self

    def _fill(self):
        """Compact the unconsumed tail to the front and read more; returns bytes added."""
        if self._eof:
            return 0
        tail = self._end - self._pos
        if self._pos:
            self._buf[:tail] = self._buf[self._pos:self._end]
            self._pos, self._end = 0, tail
        room = len(self._buf) - self._end
        if room == 0:  # a single line longer than the buffer: grow (malformed input ends up here)
            self._buf.extend(bytes(len(self._buf)))
            room = len(self._buf) - self._end
        view = memoryview(self._buf)[self._end:self._end + room]
        got = _read_arrived(self.stream, view)
        del view
        if got == 0:
            self._eof = True
        self._end += got
        return got

    def next_batch(self, max_blocks):
        """-> (timestamps, block_idx int64[n], text bytearray, payload_off int64[n]) or None at EOF.
        `text` is this reader's buffer: valid until the next call.  Lines are framed by the
        engine library's host routine (`thr_frame_card`, ~0.1 us per line); `_next_batch_py` is
        the same logic in Python for installations where the library is not built."""
        global _frame_card
        if _frame_card is None:
            try:
                from thrifty_amd import _native
                _native.load_library()
                _frame_card = _native.frame_card
            except Exception:      # no library: host-side text framing still works
                _frame_card = False
        if _frame_card is False:
            return self._next_batch_py(max_blocks)
        from thrifty_amd._native import NativeError
        while True:
            try:
                ts, idx, off, nxt = _frame_card(self._buf, self._pos, self._end, self.block_len,
                                                self._eof, max_blocks)
            except NativeError as exc:
                raise ValueError(str(exc))
            progressed = nxt > self._pos
            self._pos = nxt
            if len(off):
                return ts.tolist(), idx, self._buf, off
            if self._eof:
                if not progressed:
                    return None
                continue
            self._fill()        # (sets _eof when the stream is exhausted; the tail is framed next round)

    def _next_batch_py(self, max_blocks):
        stamps, idxs, offs = [], [], []
        buf = self._buf
        chars = self.payload_chars
        while len(offs) < max_blocks:
            # Fast path: a data line is `<ts> <idx> ` + exactly payload_chars characters, so its
            # end is known from the two spaces of the short header -- no 43 KB newline scan.
            start = self._pos
            if start < self._end and 0x30 <= buf[start] <= 0x39:
                sp1 = buf.find(b" ", start, min(start + 40, self._end))
                sp2 = buf.find(b" ", sp1 + 1, min(sp1 + 32, self._end)) if sp1 >= 0 else -1
                end = sp2 + 1 + chars
                if sp2 >= 0 and end < self._end and (
                        buf[end] == 0x0A or (buf[end] == 0x0D and end + 1 < self._end and buf[end + 1] == 0x0A)):
                    try:
                        ts, bi = float(buf[start:sp1]), int(buf[sp1 + 1:sp2])
                    except ValueError:
                        # a header like `12x.5 7 <payload>`: the records framed so far go out first (as in
                        # the general path below and in thr_frame_card); the next call raises at this line
                        if offs:
                            break
                        raise ValueError("malformed .card header: %r" % bytes(buf[start:sp2]))
                    stamps.append(ts)
                    idxs.append(bi)
                    offs.append(sp2 + 1)
                    self._pos = end + (1 if buf[end] == 0x0A else 2)
                    continue
            end = buf.find(b"\n", self._pos, self._end)
            if end < 0:
                if offs:
                    break          # never let a batch straddle a refill: offsets index ONE buffer state
                if self._fill():
                    buf = self._buf
                    continue
                if self._pos >= self._end:
                    break
                end = self._end    # last line without newline
            start, self._pos = self._pos, min(end + 1, self._end)
            stop = end - 1 if end > start and buf[end - 1] == 0x0D else end
            if stop <= start or buf[start] == 0x23:      # blank or '#'
                continue
            if buf[start:start + 19] == b"Using Volk machine:" or buf[start:start + 6] == b"linux;":
                continue
            sp1 = buf.find(b" ", start, stop)
            sp2 = buf.find(b" ", sp1 + 1, stop)
            bad = sp1 < 0 or sp2 < 0 or stop - (sp2 + 1) != self.payload_chars
            if bad and offs:
                # the records before a bad line go out first (the reference's per-line loop had
                # processed them); the next call starts at the bad line and raises
                self._pos = start
                break
            if sp1 < 0 or sp2 < 0:
                raise ValueError("malformed .card line: %r" % bytes(buf[start:min(stop, start + 60)]))
            if stop - (sp2 + 1) != self.payload_chars:
                raise ValueError("block %s: payload of %d base64 characters, expected %d (block_len %d)" % (
                    bytes(buf[sp1 + 1:sp2]).decode(), stop - (sp2 + 1), self.payload_chars, self.block_len))
            try:
                ts, bi = float(buf[start:sp1]), int(buf[sp1 + 1:sp2])
            except ValueError:
                if offs:
                    self._pos = start
                    break
                raise ValueError("malformed .card header: %r" % bytes(buf[start:sp2]))
            stamps.append(ts)
            idxs.append(bi)
            offs.append(sp2 + 1)
        if not offs:
            return None
        return stamps, np.asarray(idxs, dtype=np.int64), buf, np.asarray(offs, dtype=np.int64)

    @property
    def live(self):
        return is_live(self.stream)

    def __iter__(self):
        while True:
            batch = self.next_batch(64)
            if batch is None:
                return
            stamps, idxs, text, offs = batch
            for ts, idx, off in zip(stamps, idxs, offs):
                raw = np.frombuffer(base64.b64decode(bytes(text[off:off + self.payload_chars])), dtype=np.uint8)
                yield ts, int(idx), IQBlock(raw_to_complex(raw), raw)


class RawStream(object):
    """Batch-oriented raw u8 I/Q reader with the overlap framing done on the GPU
    (reference block_data.py:70-98, fastcard raw_reader.c:15-46).

    `block_reader` re-copies every block's history on the host; here the host keeps the
    byte stream contiguous (the last 2*history bytes are carried over between batches) and
    the engine reads overlapping windows straight out of it (`Engine.detect_stream`).  The
    first ceil(history / (size - history)) blocks still contain part of the reference's
    all-zero initial history, which has no u8 form: those come back as complex64 blocks.
    Iterating a RawStream yields `block_reader`'s tuples (drop-in use, host framing).
    """

    def __init__(self, stream, size, history):
        self.stream = stream
        self.size, self.history = int(size), int(history)
        self.new = self.size - self.history
        if self.new <= 0:
            raise ValueError("history must be shorter than the block")
        self.device_framing = self.new % 2 == 0     # 4-byte aligned block starts
        self._next_idx = 0
        self._n_lead = -(-self.history // self.new)          # blocks that still see zero history
        self._lead = np.zeros(self.size, dtype=np.complex64)  # rolling block of the lead-in
        # _buf = [carry: last 2*history stream bytes][unconsumed new bytes]; _have = valid bytes
        self._buf = bytearray(2 * self.history)
        self._have = 2 * self.history
        self._consumed = 0      # bytes of the previous u8 batch still to be slid out
        self._eof = False
        self._arrivals = []     # (valid bytes after the read, time.time()) of this batch's reads
        # regular file: overlapping blocks are plain slices of the mapping (no carry, no copy)
        self._map, self._off = _map_regular_file(stream)
        if self._map is None:
            _widen_pipe(stream)
        self._origin = self._off    # stream byte 0 (the caller may have consumed a header)
        self._stop_idx = None       # sharded mapped file: one past this rank's last block

    @property
    def mapped(self):
        """True if the input is a regular file read through mmap."""
        return self._map is not None

    def mapped_span(self):
        """The bytes of the mapped file this reader will hand out (its shard, with the history in
        front of its first block), as a memoryview; None if not mapped (see CardStream)."""
        if self._map is None:
            return None
        step, carry = 2 * self.new, 2 * self.history
        lo = max(0, self._off - carry)
        hi = len(self._map)
        if self._stop_idx is not None:
            hi = min(hi, self._off + max(0, self._stop_idx - self._next_idx) * step)
        return memoryview(self._map)[lo:hi]

    def ready(self, release=None):
        """True if next_batch() would not have to WAIT for the source: a mapped file, the end of
        the stream, or a whole block's worth of new samples buffered (see CardStream.ready; bytes
        waiting in the pipe are taken in first, after `release()`)."""
        if self._map is not None or self._eof:
            return True
        carry, step = 2 * self.history, 2 * self.new
        if self._have - self._consumed - carry >= step:
            return True
        if not _readable_within(self.stream, 0):
            return False
        if release is not None:
            release()
        self._slide(self._consumed)
        self._consumed = 0
        if len(self._buf) < carry + step:          # (room for one block at least; next_batch grows it)
            self._buf.extend(bytes(carry + step - len(self._buf)))
        if len(self._buf) > self._have:
            view = memoryview(self._buf)[self._have:]
            got = _read_arrived(self.stream, view)
            del view
            if got == 0:
                self._eof = True
            self._have += got
        return self._eof or self._have - carry >= step

    def shard(self, rank, world):
        """Restrict a mapped raw file to this rank's contiguous block range.  The lead-in blocks
        (zero initial history) stay with rank 0; the rest is split evenly."""
        if world <= 1:
            return self
        if self._map is None:
            raise ValueError("only a regular file can be sharded over GPUs (not a pipe)")
        step = 2 * self.new
        total = (len(self._map) - self._origin) // step
        lead = min(self._n_lead, total)
        base, rem = divmod(total - lead, world)
        lo = lead + rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        if rank == 0:
            lo = 0
        else:
            self._next_idx = lo
            self._off = self._origin + lo * step
        self._stop_idx = hi
        return self

    def take_rest(self):
        """Mapped file only, after the zero-history lead-in (`in_lead_in` false): every block not
        handed out yet as ONE u8 stream view whose first 2 * size bytes are block `first` -- for a
        consumer that frames it itself (thr_run_stream).  -> (view, first, n_blocks); the reader is
        exhausted afterwards."""
        if self._map is None or self.in_lead_in:
            raise ValueError("take_rest() needs a mapped regular file behind its lead-in blocks")
        step, carry = 2 * self.new, 2 * self.history
        n = (len(self._map) - self._off) // step
        if self._stop_idx is not None:
            n = min(n, self._stop_idx - self._next_idx)
        n = max(0, n)
        first = self._next_idx
        view = memoryview(self._map)[self._off - carry:self._off + n * step] if n else memoryview(b"")
        self._next_idx += n
        self._off += n * step
        return view, first, n

    @property
    def in_lead_in(self):
        """The next block still contains part of the reference's all-zero initial history."""
        return self._next_idx < self._n_lead

    def _read_upto(self, want_end, need_end=None):
        """Read until `need_end` valid bytes are buffered (default: want_end) or EOF, never
        asking for more than `want_end`: on a live pipe a batch is whatever has arrived, as
        long as it holds at least one block."""
        need_end = want_end if need_end is None else need_end
        if len(self._buf) < want_end:
            self._buf.extend(bytes(want_end - len(self._buf)))
        # (bytes already waiting in the pipe are taken too, even if ready()'s top-up has satisfied
        # `need_end` before this call: a batch is what HAS arrived, not the first block of it)
        while not self._eof and (self._have < need_end or
                                 (self._have < want_end and _readable_within(self.stream, 0))):
            view = memoryview(self._buf)[self._have:want_end]
            got = _read_arrived(self.stream, view)
            del view
            if got == 0:
                self._eof = True
            self._have += got
            self._arrivals.append((self._have, time.time()))

    def _stamps(self, ends):
        """Arrival time of each block's last byte (the reference's block_reader stamps a block
        when its read returns, block_data.py:86-98): `ends` = buffer offsets one past each block."""
        if not self._arrivals:      # everything was already buffered
            return [time.time()] * len(ends)
        out, k = [], 0
        for e in ends:
            while k + 1 < len(self._arrivals) and self._arrivals[k][0] < e:
                k += 1
            out.append(self._arrivals[k][1])
        return out

    def _slide(self, used):
        """Drop `used` consumed bytes: the bytes before the new position become the carry."""
        if used:
            tail = self._have - used
            self._buf[:tail] = self._buf[used:self._have]
            self._have = tail

    def next_batch(self, max_blocks):
        """-> ("c64", stamps, idxs int64[n], complex64[n, size])  for the lead-in blocks,
              ("u8", stamps, idxs int64[n], memoryview of the bytes of n overlapping blocks),
              or None at EOF.  The u8 view aliases this reader's buffer: valid until the next call."""
        step, carry = 2 * self.new, 2 * self.history
        if self._map is not None:
            return self._next_batch_mapped(max_blocks, step, carry)
        self._slide(self._consumed)
        self._consumed = 0
        if self._next_idx < self._n_lead:
            blocks, idxs, stamps = [], [], []
            while len(blocks) < max_blocks and self._next_idx < self._n_lead:
                self._arrivals = []
                self._read_upto(carry + step)
                if self._have < carry + step:
                    break                                   # short tail: dropped, like the reference
                stamps.append(time.time())
                chunk = np.frombuffer(self._buf, dtype=np.uint8, count=step, offset=carry)
                self._lead = np.concatenate([self._lead[self.new:], raw_to_complex(chunk)])
                del chunk
                blocks.append(self._lead)
                idxs.append(self._next_idx)
                self._next_idx += 1
                self._slide(step)
            if not blocks:
                return None
            return "c64", stamps, np.asarray(idxs, dtype=np.int64), np.stack(blocks)
        self._arrivals = []
        self._read_upto(carry + max_blocks * step, need_end=carry + step)
        n = (self._have - carry) // step
        if n <= 0:
            return None
        idxs = np.arange(self._next_idx, self._next_idx + n, dtype=np.int64)
        self._next_idx += n
        self._consumed = n * step
        stamps = self._stamps([carry + (i + 1) * step for i in range(n)])
        return "u8", stamps, idxs, memoryview(self._buf)[:carry + n * step]

    def _next_batch_mapped(self, max_blocks, step, carry):
        m, size = self._map, len(self._map)
        if self._next_idx < self._n_lead:
            blocks, idxs = [], []
            while (len(blocks) < max_blocks and self._next_idx < self._n_lead and self._off + step <= size
                   and (self._stop_idx is None or self._next_idx < self._stop_idx)):
                chunk = np.frombuffer(m, dtype=np.uint8, count=step, offset=self._off)
                self._lead = np.concatenate([self._lead[self.new:], raw_to_complex(chunk)])
                blocks.append(self._lead)
                idxs.append(self._next_idx)
                self._next_idx += 1
                self._off += step
            if not blocks:
                return None
            return "c64", [time.time()] * len(blocks), np.asarray(idxs, dtype=np.int64), np.stack(blocks)
        n = min(max_blocks, (size - self._off) // step)
        if self._stop_idx is not None:
            n = min(n, self._stop_idx - self._next_idx)
        if n <= 0:
            return None
        # block i of the batch starts `carry` bytes before its new samples; the lead-in has
        # consumed at least `carry` bytes, so the slice never reaches before the stream's start
        view = memoryview(m)[self._off - carry:self._off + n * step]
        idxs = np.arange(self._next_idx, self._next_idx + n, dtype=np.int64)
        self._next_idx += n
        self._off += n * step
        return "u8", [time.time()] * n, idxs, view

    @property
    def live(self):
        return is_live(self.stream)

    def __iter__(self):
        return block_reader(self.stream, self.size, self.history)


def card_line(timestamp, block_idx, raw):
    """Format one .card line (fastcard_cli.c:187-192: "%ld.%06ld %PRId64 %s\\n")."""
    sec, usec = divmod(int(round(timestamp * 1e6)), 1000000)   # (a fraction >= .9999995 carries)
    return "%d.%06d %d %s\n" % (sec, usec, block_idx,
                                base64.b64encode(np.asarray(raw, np.uint8).tobytes()).decode())
