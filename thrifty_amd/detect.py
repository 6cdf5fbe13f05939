"""Detect positioning signals and estimate sample-of-arrival -- on an MI355X.

Drop-in for the reference's operator API (thrifty/detect.py): same
`DetectorSettings` tuple, same `Detector(settings, blocks=None, rxid=-1,
yield_data=False)` constructor, `.detect()` / iterator protocol, result types and
`.toad` text, same `detector_cli` factory hook.  The per-block work
(carrier_sync.py + soa_estimator.py in the reference) runs in the HIP engine
behind include/thrifty_hip.h; this module only batches blocks, calls the C ABI
and re-hydrates records in input order.  There is no CPU fallback.
"""
from __future__ import annotations

import logging
import time
from collections import deque, namedtuple
from types import SimpleNamespace

import numpy as np

from thrifty_amd import _native, toads_data
from thrifty_amd.stages import SoaStage, SyncStage, unique_window      # noqa: F401  (unique_window: part of this module's API)
try:
    from thrifty_amd import _fastresults
except ImportError as _exc:      # pragma: no cover -- a tree that was never built
    raise ImportError("thrifty_amd._fastresults is not built: run `python -m thrifty_amd.build` (%s)" % _exc)
from thrifty_amd.block_data import CardStream, RawStream

DetectorSettings = namedtuple("DetectorSettings", [
    "block_len", "history_len", "carrier_len", "carrier_thresh", "carrier_window",
    "template", "corr_thresh"])


_LOG = logging.getLogger("thrifty_amd.detect")

_SLOW_SOURCE_S = 0.002   # inter-arrival time above which a LIVE source ends the batch being filled
_UNKNOWN_FILL_S = 0.05   # a source that does not say whether it is live: longest time spent FILLING one batch
_YIELD_DATA_BATCH = 64    # blocks per batch with yield_data (each drags two N-point dumps along)


def _offset_mode(offset_type):
    """thr_format_toad's carrier_offset_f32 argument for a Detector's `_offset_type`."""
    return 0 if offset_type is float else 2 if offset_type is int else 1


class _Deferred(object):
    """An exception that belongs to one position of a batch: raised when that position is
    reached, after the results before it have been delivered."""

    def __init__(self, exc):
        self.exc = exc


class Detector(object):
    """All-in-one carrier sync + matched filter + SoA estimator, batched on the GPU.

    Parameters are the reference's (detect.py:40); `batch_size` and `device_id`
    are additions: up to `batch_size` blocks are pulled from `blocks` and
    processed per launch, results are handed out one per input block, in order.
    """

    _fit_reach = 3            # the carrier interpolator reads fft_mag[peak + 3] (carrier_sync.py:187)
    _offset_type = float      # CarrierSyncInfo.offset as the reference types it
    _multi = False            # MultiTemplateDetector: settings.template is [n_templates, len]
    _host_interp = False      # `sync.interpolator` has been assigned: the two-pass slow path
    _host_soa = False         # `soa_estimate.interpolate` has been assigned: the correlation comes to the host
    strict_fit = False        # True: a block flagged THR_FLAG_FIT_UNCONVERGED ends the iteration with RuntimeError

    @property
    def _host_path(self):
        """A stage has been replaced by a host callable: batches run one at a time through
        _detect_batch_host (no tickets, no device ingest, no library loop)."""
        return self._host_interp or self._host_soa

    def __init__(self, settings, blocks=None, rxid=-1, yield_data=False, batch_size=None,
                 device_id=0, _preshift_num=0, _fastdet=False, max_wait=None, max_fill=None,
                 pin_input=True, _interpolator="parabolic", _path="auto", populate_threads=0, low_cpu=False,
                 strict_fit=False):
        """Batching a classic `(timestamp, idx, block)` iterator must not hold results back the way
        the reference's per-block loop never did.  What ends the batch being filled (what has
        arrived is processed instead of waiting for a full batch) depends on what the source says
        about itself through a `.live` attribute (block_data's readers have one; a wrapper around
        them should pass it on):
          * `.live` true (a reader over a pipe / socket / tty): a next() that takes longer than
            `max_wait` (default 2 ms) ends the batch, and the following batch is never read ahead;
          * `.live` false (a regular file, an in-memory list): no limit, batches fill up and the
            next one is read while this one's results are handed out;
          * no `.live` at all (any user generator, filter or socket reader): a batch is filled for
            at most `max_fill` seconds (default 50 ms) and never read ahead -- a CPU-bound source
            (host decode, gzip) still gets batches of hundreds of blocks, a live one a latency of
            50 ms instead of batch_size blocks' worth of waiting.
        `max_wait` / `max_fill` given explicitly apply to any source.  `pin_input`: page-lock a
        mapped input file ahead of the copies while it is read (thr_input_window; best effort);
        `populate_threads`: the library threads that map its pages ahead of the locking (0 = the
        library's default; a rank of a sharded run passes parallel.populate_threads(world));
        `low_cpu`: wait for batches by asking and napping instead of polling (thr_set_wait_mode: about
        half a CPU less per detector, what a rank takes when the node's CPUs are short).
        `strict_fit`: the reference's loop dies with RuntimeError("Optimal parameters not found: ...")
        on a block whose Dirichlet fit SciPy's curve_fit gives up on (carrier_sync.py:189, uncaught --
        degenerate geometries only: templates far shorter than block_len / 24).  The engine's fit is
        the same MINPACK routine and flags such a block (THR_FLAG_FIT_UNCONVERGED); by default the
        block still gets its record (from the fit's last iterate), with strict_fit=True the iteration
        ends there with RuntimeError after the results before it, like the reference's.  Whether the
        fit runs out of evaluations hangs on the last digit of seven float32 magnitudes, so the flagged
        blocks are the reference's up to that noise -- not guaranteed block for block."""
        self.strict_fit = bool(strict_fit)
        if batch_size is None:      # ~64 MiB of u8 samples per engine batch (the staging chunk size)
            batch_size = max(64, min(65536, (64 << 20) // (2 * int(settings.block_len))))
            if yield_data:          # every block of a batch holds two N-point stage dumps in _ready
                batch_size = min(batch_size, _YIELD_DATA_BATCH)
        live = getattr(blocks, "live", None)
        if live is None and isinstance(blocks, (list, tuple, np.ndarray)):
            live = False            # an in-memory sequence never has to be waited for
        self._known_not_live = live is not None and not live
        if max_wait is None:
            max_wait = _SLOW_SOURCE_S if live else float("inf")
        if max_fill is None:
            max_fill = _UNKNOWN_FILL_S if live is None else float("inf")
        self.max_wait, self.max_fill = float(max_wait), float(max_fill)
        self.settings = settings
        # a CardStream is consumed in whole batches with the base64 payloads decoded on the GPU
        self._card = blocks if isinstance(blocks, CardStream) and not yield_data else None
        # a RawStream likewise: the overlapping blocks are framed on the GPU from the byte stream
        self._raw = (blocks if isinstance(blocks, RawStream) and blocks.device_framing
                     and not yield_data else None)
        self.blocks = iter(blocks) if blocks is not None else None
        self.rxid = rxid
        self.yield_data = yield_data
        self.batch_size = max(1, int(batch_size))
        self.new_len = settings.block_len - settings.history_len
        template = np.asarray(settings.template)
        if template.ndim != (2 if self._multi else 1):
            raise ValueError("Detector takes one 1-D template, MultiTemplateDetector a "
                             "[n_templates, template_len] array")
        self._engine = _native.Engine(
            settings.block_len, settings.history_len, template, settings.carrier_thresh,
            settings.carrier_window, settings.corr_thresh, carrier_len=settings.carrier_len,
            device_id=device_id, max_batch=self.batch_size, preshift_num=_preshift_num,
            fastdet=_fastdet, interpolator=_interpolator, path=_path)
        if low_cpu:
            self._engine.set_wait_mode(True)
        # which kernels this handle's launches take, and why (thr_get_path_info): a stddev term, a
        # long template or an odd history silently costs the sectioned correlate stage -- say so once
        self.engine_path = self._engine.path_info()
        _LOG.info("engine path: %s", self.engine_path["text"])
        # a mapped input file becomes the engine's input window: a library thread page-locks it a
        # bounded distance ahead of the chunk copies, which are then asynchronous DMA out of the
        # page cache -- this thread frames the next batch and formats the previous one meanwhile
        self._pin = False
        reader = self._card if self._card is not None else self._raw
        span = reader.mapped_span() if reader is not None else None
        if pin_input and span is not None and len(span):
            self._engine.input_window(span, populate_threads=populate_threads)
            self._pin = True
        self._ready = deque()
        self._exhausted = False
        # submitted batches whose records have not been collected yet, oldest first: one ahead of
        # the batch being handed out, two when the input is a page-locked file (the copies are
        # asynchronous then, and the DMA engine stays busy while this thread formats)
        self._ahead = deque()
        self._depth = 2 if self._pin else 1
        self._read_error = None     # an exception the read-ahead hit: raised once the batch before it is out
        # batch readers only (CardStream / RawStream): hand out detections only, skipping the
        # per-block Python objects of everything else (set by detector_cli under --quiet)
        self.only_detections = False
        corr_len = settings.block_len - template.shape[-1] + 1
        # twins of the reference's sub-objects (detect.py:46-58): the same attributes, and CALLABLE
        # like them -- evaluated by the engine, one block at a time (_SyncStage / _SoaStage below)
        self._host_interp = self._host_soa = False   # a stage replaced by a host callable: the slow path below
        self.sync = SyncStage(self, settings)
        self.soa_estimate = SoaStage(self, settings, template, corr_len)

    # ------------------------------------------------------------------ core
    def _stack(self, blocks):
        """-> (array, is_u8).  u8 fast path only if every block still has its raw bytes."""
        n = self.settings.block_len
        raws = [getattr(b, "raw", None) if not (isinstance(b, np.ndarray) and b.dtype == np.uint8)
                else b for b in blocks]
        if all(r is not None and len(r) == 2 * n for r in raws):
            return np.stack([np.asarray(r, dtype=np.uint8) for r in raws])
        for b in blocks:
            assert len(b) == n
        return np.stack([np.asarray(b).astype(np.complex64) for b in blocks])

    def _result(self, timestamp, block_idx, rec):
        flags = int(rec["flags"])
        if flags & _native.FLAG_INDEX_ERROR:
            n = self.settings.block_len
            # the reference indexes fft_mag[peak_idx + reach] without wrapping (carrier_sync.py:187)
            raise IndexError("index {} is out of bounds for axis 0 with size {}".format(
                max(int(rec["carrier_bin"]) + self._fit_reach, n), n))
        if self.strict_fit and flags & _native.FLAG_FIT_UNCONVERGED:
            raise self._fit_error()
        has_carrier = bool(flags & _native.FLAG_CARRIER)
        carrier = toads_data.CarrierSyncInfo(
            int(rec["carrier_bin"]),
            (0 if flags & _native.FLAG_INT_OFFSET else self._offset_type(rec["carrier_offset"])) if has_carrier else 0,
            np.float32(rec["carrier_energy"]), np.float32(rec["carrier_noise"]))
        if not has_carrier:
            return False, toads_data.DetectionResult(timestamp, block_idx, None, carrier, None,
                                                     self.rxid)
        detected = bool(flags & _native.FLAG_CORR)
        corr = toads_data.CorrDetectionInfo(
            int(rec["corr_sample"]), float(rec["corr_offset"]) if detected else 0,
            float(rec["corr_energy"]), float(rec["corr_noise"]))
        soa = self.new_len * block_idx + corr.sample + corr.offset
        return detected, toads_data.DetectionResult(timestamp, block_idx, soa, carrier, corr,
                                                    self.rxid)

    def _results(self, stamps, idxs, recs):
        """Vectorised re-hydration of a batch of records (one .tolist() per column instead of
        ten numpy-scalar look-ups per block).  A block on which the reference raises IndexError
        (carrier_sync.py:187) ends the list with a `_Deferred` exception: the results of the
        blocks before it are still handed out, as the reference's per-block loop does."""
        flags = recs["flags"]
        bad = np.flatnonzero(flags & self._fatal_flags)
        if len(bad):
            k = int(bad[0])
            out = self._results(stamps[:k], idxs[:k], recs[:k])
            try:
                self._result(stamps[k], int(idxs[k]), recs[k])
            except (IndexError, RuntimeError) as exc:
                out.append(_Deferred(exc))
            return out
        # one C call per batch (csrc/fastresults.c): a result keeps its 64-byte record and makes each
        # attribute -- the namedtuples, soa, the np.float32 energies -- when it is first read
        recs = np.ascontiguousarray(recs)
        if not np.array_equal(recs["block_idx"], idxs):     # (the engine echoes the index it was given)
            recs = recs.copy()
            recs["block_idx"] = idxs
        return _fastresults.build(self._result_context(), recs, stamps)

    def _result_context(self):
        ctx = getattr(self, "_ctx", None)
        key = (self.new_len, self.rxid, self._offset_type, self._multi)
        if ctx is None or ctx[0] != key:
            ctx = self._ctx = (key, _fastresults.Context(
                int(self.new_len), self.rxid, self._offset_type, np.float32, toads_data.CarrierSyncInfo,
                toads_data.CorrDetectionInfo, toads_data.DetectionResult, bool(self._multi),
                _native.format_toad_address(), _offset_mode(self._offset_type)))
        return ctx[1]

    @property
    def _fatal_flags(self):
        """Record flags on which the reference's per-block loop dies (the iteration ends there)."""
        return _native.FLAG_INDEX_ERROR | (_native.FLAG_FIT_UNCONVERGED if self.strict_fit else 0)

    @staticmethod
    def _fit_error():
        # scipy/optimize/_minpack_py.py: `raise RuntimeError("Optimal parameters not found: " + errmsg)`
        return RuntimeError("Optimal parameters not found: the Dirichlet carrier fit did not converge "
                            "(MINPACK lmdif exit code 5..8)")

    def _run(self, arr, idx):
        """Records [B, n_templates] of the blocks in `arr`, now.  While the iterator has a batch
        in flight the synchronous entry point would refuse to run (thr_detect waits for open
        tickets to be collected), so a direct detect() between two next() calls rides the
        ticket interface too."""
        if not self._ahead or self.yield_data or self._host_path:
            return self._engine.detect(arr, idx)
        step = self.batch_size
        return np.concatenate([self._engine.collect(self._engine.submit(arr[s:s + step], idx[s:s + step]))
                               for s in range(0, len(arr), step)])

    # --------------------------------------------- a replaced carrier interpolator (slow path)
    def _use_host_interpolator(self):
        """`sync.interpolator = fn` was assigned (reference experimental/detect_carrier_interpol.py:
        17-40): from here on every batch makes two engine passes with the callable in between, one
        batch at a time, blocks pulled through the classic per-block iterator."""
        self._enter_host_path("sync.interpolator")
        self._host_interp = True

    def _use_host_soa_interpolator(self):
        """`soa_estimate.interpolate = fn` was assigned (reference experimental/detect_xcorr_interpol.py:
        36-62): the engine's verdicts and peak stay, the correlation magnitudes of the detected blocks
        come to the host (stage dump) and the callable `(corr_mag, peak_idx) -> offset` replaces the
        log-parabola; the offset is clipped to +-0.6 like the reference's (soa_estimator.py:87-89)."""
        self._enter_host_path("soa_estimate.interpolate")
        self._host_soa = True

    def _enter_host_path(self, what):
        if self._engine.preshift_num or self._multi or self.yield_data:
            raise NotImplementedError("a replaced %s is offered by the default single-template Detector "
                                      "(the variants interpolate inside their fused kernels)" % what)
        if self._ahead or self._ready:
            raise RuntimeError("assign %s before iterating" % what)
        self._card = self._raw = None          # (the readers also iterate as (timestamp, idx, block))
        if self._pin:
            self._engine.input_window(None)
            self._pin = False
        self.batch_size = min(self.batch_size, _YIELD_DATA_BATCH)   # a batch drags its spectra along

    def _detect_batch_host(self, items):
        """Reference Synchronizer.sync with `interpolator` replaced (carrier_sync.py:52-76), for a batch:
        engine pass 1 -> verdict, peak bin and |FFT#1| of every block; the callable on the host for the
        blocks that carry a carrier; engine pass 2 with those offsets (thr_detect_offsets).  An
        exception of the callable (IndexError at the spectrum's end, like the reference's own
        interpolators) belongs to its block: the results before it still come out, then it is raised."""
        arr = self._stack([it[2] for it in items])
        idx = np.array([int(it[1]) for it in items], dtype=np.int64)
        vals, error, n_ok, offsets = [0] * len(items), None, len(items), None
        if self._host_interp:
            fn = self.sync._interpolator
            first = self._engine.detect(arr, idx)[:, 0]
            has = (first["flags"] & (_native.FLAG_CARRIER | _native.FLAG_INDEX_ERROR)) != 0
            spectra = self._engine.debug_fft(arr) if has.any() and fn is not None else None
            for i in np.flatnonzero(has).tolist():
                if fn is None:
                    continue                        # `if self.interpolator is not None` (carrier_sync.py:66)
                try:
                    vals[i] = fn(np.abs(spectra[i]), int(first["carrier_bin"][i]))
                except Exception as exc:            # noqa: BLE001 -- whatever the callable raises is the block's
                    error, n_ok = exc, i
                    break
            offsets = np.array([float(v) for v in vals[:n_ok]])
            wild = np.flatnonzero(~np.isfinite(offsets))
            if len(wild):       # the reference's shifter raises on such a block (int(round(nan)), carrier_sync.py:241-245)
                n_ok = int(wild[0])
                error = ValueError("sync.interpolator returned %r for block %d: the carrier offset is not finite"
                                   % (vals[n_ok], int(idx[n_ok])))
                offsets = offsets[:n_ok]
        out = []
        if n_ok:
            recs = (self._engine.detect_offsets(arr[:n_ok], offsets, idx[:n_ok]) if offsets is not None
                    else self._engine.detect(arr[:n_ok], idx[:n_ok]))[:, 0]
            out = self._results([it[0] for it in items[:n_ok]], idx[:n_ok], recs)
            if offsets is not None:
                for (detected, res), v in zip(out, vals):
                    if not isinstance(res, _Deferred) and res.corr_info is not None:
                        # the offset as the callable returned it (none() -> the int 0)
                        res.carrier_info = res.carrier_info._replace(offset=v)
            if self._host_soa:
                out, err2 = self._interpolate_on_host(out, arr[:n_ok], recs, offsets)
                error = err2 if err2 is not None else error
        if error is not None:
            out.append(_Deferred(error))
        return out

    def _interpolate_on_host(self, out, arr, recs, offsets):
        """`soa_estimate.interpolate` replaced: for every DETECTED block the callable gets the
        correlation magnitudes (the reference's `corr.mag`, all corr_len lags) and the peak index
        (soa_estimator.py:87: `offset = 0 if not detected else self.interpolate(corr.mag, peak_idx)`),
        its value is clipped to +-0.6 and becomes CorrDetectionInfo.offset; `soa` follows.
        `soa_estimate.last_fft` holds the block's shifted spectrum meanwhile (what the reference's
        IterativeSoaEstimator keeps, detect_xcorr_interpol.py:27-34).  -> (results up to an exception
        of the callable, that exception or None)"""
        fn = self.soa_estimate._interpolate
        hits = [i for i, item in enumerate(out) if not isinstance(item, _Deferred) and item[0]]
        if not hits:
            return out, None
        xhat, corr = self._engine.debug_stage(arr, carrier_offset=offsets)
        corr_len = self.soa_estimate.corr_len
        for i in hits:
            detected, res = out[i]
            self.soa_estimate.last_fft = xhat[i]
            try:
                off = fn(np.abs(corr[i][:corr_len]), res.corr_info.sample)
            except Exception as exc:                # noqa: BLE001 -- the block's own error
                return out[:i], exc
            off = -0.6 if off < -0.6 else 0.6 if off > 0.6 else off       # _clip_offset (soa_estimator.py:16-17)
            res.corr_info = res.corr_info._replace(offset=off)
            res.soa = self.new_len * res.block + res.corr_info.sample + off
        self.soa_estimate.last_fft = None
        return out, None

    def detect_batch(self, items):
        """[(timestamp, block_idx, block), ...] -> [(detected, DetectionResult), ...]."""
        if not items:
            return []
        if self._host_path:
            out = self._detect_batch_host(items)
            if out and isinstance(out[-1], _Deferred):
                raise out[-1].exc
            return out
        arr = self._stack([it[2] for it in items])
        idx = np.array([int(it[1]) for it in items], dtype=np.int64)
        recs = self._run(arr, idx)[:, 0]
        out = self._results([it[0] for it in items], idx, recs)
        if out and isinstance(out[-1], _Deferred):
            raise out[-1].exc
        return out

    def detect(self, timestamp, block_idx, block):
        """Process one block (reference detect.py:60-78)."""
        assert len(block) == self.settings.block_len or (
            getattr(block, "dtype", None) == np.uint8 and len(block) == 2 * self.settings.block_len)
        detected, result = self.detect_batch([(timestamp, block_idx, block)])[0]
        if not self.yield_data:
            return detected, result
        shifted_fft = corr = None
        if result.corr_info is not None:
            xhat, cc = self._engine.debug_stage(self._stack([block]))
            shifted_fft, corr = xhat[0], cc[0][:self.soa_estimate.corr_len]
        return detected, result, shifted_fft, corr

    # -------------------------------------------------------------- iterator
    def _flat(self, stamps, idxs, recs):
        """Engine records [B, n_templates] -> one record per result slot (this class: template 0)."""
        return stamps, idxs, recs[:, 0]

    def _submit_next(self, prev=None):
        """Pull one batch from the block source and hand it to the engine WITHOUT waiting for it
        (thr_submit*): -> an opaque pending batch, or None when the source is exhausted.  `prev`:
        the ticket still in flight -- a reader that refills ONE buffer (a pipe) must not touch it
        before that batch's host-to-device copies are done (a mapped file is never overwritten)."""
        reader = self._card if self._card is not None else self._raw
        if prev is not None and reader is not None and not reader.mapped:
            self._engine.inputs_consumed(prev)      # (again after _may_read_ahead's: returns at once)
        if self._card is not None:
            batch = self._card.next_batch(self.batch_size)
            if batch is None:
                self._exhausted = True
                return None
            stamps, idxs, text, offs = batch
            return stamps, idxs, self._engine.submit_card(text, offs, idxs)
        if self._raw is not None:
            batch = self._raw.next_batch(self.batch_size)
            if batch is None:
                self._exhausted = True
                return None
            kind, stamps, idxs, data = batch
            if kind == "u8":
                return stamps, idxs, self._engine.submit_stream(data, int(idxs[0]))
            # lead-in blocks that still contain the all-zero initial history
            return stamps, idxs, self._engine.submit(data, idxs)
        items = []
        t_fill = time.perf_counter()
        while len(items) < self.batch_size and not self._exhausted:
            t0 = time.perf_counter()
            try:
                items.append(next(self.blocks))
            except StopIteration:
                self._exhausted = True
            except Exception as exc:
                if not items:
                    raise
                # the blocks read before the bad one are still processed and handed out (the
                # reference's loop had emitted them); the error follows them
                self._read_error, self._exhausted = exc, True
            # a live source slower than ~500 blocks/s (a receiver delivers ~200) gains nothing
            # from batching: process what has arrived instead of waiting for a full batch
            now = time.perf_counter()
            if now - t0 > self.max_wait or now - t_fill > self.max_fill:
                break
        if not items:
            return None
        if self.yield_data or self._host_path:
            return items
        arr = self._stack([it[2] for it in items])
        idx = np.array([int(it[1]) for it in items], dtype=np.int64)
        return [it[0] for it in items], idx, self._engine.submit(arr, idx)

    def _next_records(self):
        """-> (stamps, idxs, recs) of the next batch in input order, or None at the end.  The
        batch after it is submitted BEFORE this one is waited for, so the device (and the H2D
        staging of the next inputs) works while the caller formats what it was handed."""
        if not self._ahead and self._read_error is not None:
            exc, self._read_error = self._read_error, None
            self._exhausted = True
            raise exc
        if not self._ahead:
            first = self._submit_next()
            if first is None:
                return None
            self._ahead.append(first)
        if self.yield_data or self._host_path:
            return self._ahead.popleft()
        while (len(self._ahead) <= self._depth and not self._exhausted and self._read_error is None
               and self._may_read_ahead(self._ahead[-1][2])):
            try:
                nxt = self._submit_next(self._ahead[-1][2])
            except Exception as exc:
                # the NEXT batch is unreadable (a malformed line, an engine error): the batches
                # before it still go out first, as the reference's per-line loop would have emitted
                # everything before the bad input; the error is raised once they are out
                self._read_error = exc
                break
            if nxt is None:
                break
            self._ahead.append(nxt)
        cur = self._ahead.popleft()
        stamps, idxs, ticket = cur
        return self._flat(stamps, idxs, self._engine.collect(ticket))

    @property
    def _in_flight(self):
        """The oldest submitted batch not collected yet (None: nothing in flight)."""
        return self._ahead[0] if self._ahead else None

    def _drop_ahead(self):
        """The iteration ends here (the reference's loop died on this block): never leave a ticket open."""
        while self._ahead:
            pending = self._ahead.popleft()
            if not (self.yield_data or self._host_path):
                self._engine.collect(pending[2])
        # nothing of the input is read any more: an error the read-ahead had parked belongs to
        # blocks behind the one that ended the iteration, and the input's pages can be unlocked
        self._read_error = None
        if getattr(self, "_pin", False):
            self._engine.input_window(None)
            self._pin = False

    def _may_read_ahead(self, prev=None):
        """Reading the NEXT batch before handing out this one must not delay it: fine on files and
        on readers that hold a whole record already, not on a source that may have to be waited
        for -- a live one, or one that does not say (`.live` absent).  `prev`: the ticket in
        flight; a reader that tops up its ONE buffer waits for that batch's copies first."""
        reader = self._card if self._card is not None else self._raw
        if reader is not None:
            return reader.ready(release=None if prev is None or reader.mapped
                                else (lambda: self._engine.inputs_consumed(prev)))
        return self._known_not_live

    def _package(self, results, groups):
        """Flat per-record results -> the items next() hands out (this class: as they are)."""
        return results

    def _refill(self):
        got = self._next_records()
        if got is None:
            if not self._ahead:
                self._exhausted = True
            return
        if self.yield_data:
            self._ready.extend(self.detect(*it) for it in got)
            return
        if self._host_path:
            self._ready.extend(self._detect_batch_host(got))
            return
        stamps, idxs, recs = got
        groups = None
        if self.only_detections:
            keep = np.flatnonzero(recs["flags"] & (_native.FLAG_CORR | self._fatal_flags))
            if len(keep) != len(recs):
                stamps, idxs, recs, groups = [stamps[i] for i in keep], idxs[keep], recs[keep], keep
        self._ready.extend(self._package(self._results(stamps, idxs, recs), groups))
        self._recycle(got[2])        # (the results hold their own copies of the records)

    def _recycle(self, recs):
        """A collected record array nobody reads any more goes back to the engine's pool (Engine.recycle)."""
        give = getattr(self._engine, "recycle", None)
        if give is not None:
            give(recs)

    def _more(self):
        more = not self._exhausted or bool(self._ahead) or self._read_error is not None
        if not more and getattr(self, "_pin", False):
            self._engine.input_window(None)     # every batch has been collected: unlock the input's pages
            self._pin = False
        return more

    def iter_detected_records(self):
        """Batches of (timestamps float64[k], records[k]) of the DETECTED blocks only, in input
        order -- the engine's records as they are, no per-block Python objects.  A block on
        which the reference raises IndexError (carrier_sync.py:187) ends the iteration with
        that error after the detections before it have been yielded."""
        if self.blocks is None:
            raise TypeError("Detector was constructed without a block source")
        if self.yield_data or self._host_path:
            raise TypeError("record iteration is not available with yield_data / a replaced interpolator: "
                            "iterate the detector")
        while self._more():
            got = self._next_records()
            if got is None:
                continue
            stamps, idxs, recs = got
            bad = np.flatnonzero(recs["flags"] & self._fatal_flags)
            stop = int(bad[0]) if len(bad) else len(recs)
            keep = np.flatnonzero(recs["flags"][:stop] & _native.FLAG_CORR)
            if len(keep):
                yield np.asarray(stamps, dtype=np.float64)[keep], recs[keep]      # (copies)
            if len(bad):
                bad_stamp, bad_idx, bad_rec = stamps[stop], int(idxs[stop]), recs[stop].copy()
            self._recycle(recs)
            if len(bad):
                self._exhausted = True
                self._drop_ahead()
                self._result(bad_stamp, bad_idx, bad_rec)   # raises

    # ---------------------------------------------------- the loop inside the library
    def _library_loop_ready(self):
        """True if the REST of the input can be handed to thr_run_card / thr_run_stream in one call:
        a mapped file behind a batch reader, nothing in flight or handed out yet."""
        reader = self._card if self._card is not None else self._raw
        # (strict_fit ends the run on a flag the library loop does not look at: the Python loop then)
        return (reader is not None and reader.mapped and not self.yield_data and not self._ahead
                and not self._ready and not self._exhausted and self._read_error is None
                and not self.strict_fit)

    def _index_error(self, carrier_bin):
        n = self.settings.block_len
        return IndexError("index {} is out of bounds for axis 0 with size {}".format(
            max(int(carrier_bin) + self._fit_reach, n), n))

    def _run_library_loop(self, out_fd=None, want_records=False, partial=None):
        """The rest of a mapped input through thr_run_card / thr_run_stream (framing, submission,
        collection on this thread inside the library; .toad text formatted and written by a library
        thread -- no Python per batch).  -> (stats, records or None).  The reference's IndexError
        block (carrier_sync.py:187) raises IndexError here after the detections before it are out
        (`partial`: a list that receives the record chunks as they complete, so a caller that
        catches the error still holds the records before the block)."""
        eng, mode = self._engine, _offset_mode(self._offset_type)
        n_t = eng.n_templates
        stats_all, recs_all = [], ([] if partial is None else partial)

        def run(call, cap, **kw):
            rec = np.zeros(cap * n_t, dtype=_native.RECORD_DTYPE) if want_records else None
            try:
                st = call(out_fd=out_fd, rxid=self.rxid, with_txid=self._multi, carrier_offset_mode=mode,
                          batch_blocks=self.batch_size, rec_out=rec, **kw)
            except _native.NativeError as exc:
                self._finish_library_loop()
                if rec is not None and len(exc.args) > 2:
                    recs_all.append(rec[:exc.args[2]["detections"]])
                if exc.code == _native.ERR_ARG and "not valid base64" not in str(exc):
                    raise ValueError(str(exc))        # (a malformed line: what CardStream raises)
                raise
            # (where the input window's threads and the chunk submissions spent their time)
            st["window"], st["submit_phases"] = eng.debug_window_times(), eng.debug_pipe_times()
            stats_all.append(st)
            if rec is not None:
                recs_all.append(rec[:st["detections"]])
            if st["index_error"]:
                self._finish_library_loop()
                raise self._index_error(st["index_error_bin"])

        if self._card is not None:
            view, cap = self._card.take_rest()
            run(lambda **kw: eng.run_card(view, **kw), cap)
        else:
            raw = self._raw
            while raw.in_lead_in:               # the zero-history lead-in blocks (complex64, one batch)
                batch = raw.next_batch(self.batch_size)
                if batch is None:
                    break
                _, stamps, idxs, data = batch
                recs = eng.detect(data, idxs).reshape(-1)
                ts = np.repeat(np.asarray(stamps, dtype=np.float64), n_t)
                bad = np.flatnonzero(recs["flags"] & _native.FLAG_INDEX_ERROR)
                stop = int(bad[0]) if len(bad) else len(recs)
                keep = np.flatnonzero(recs["flags"][:stop] & _native.FLAG_CORR)
                if len(keep) and out_fd is not None:
                    _write_fd(out_fd, _native.format_toad(recs[keep], ts[keep], self.new_len, rxid=self.rxid,
                                                          with_txid=self._multi, carrier_offset_f32=mode))
                if len(keep) and want_records:
                    r = recs[keep].copy()
                    r["reserved"] = ts[keep].view(np.uint64)
                    recs_all.append(r)
                stats_all.append({"blocks": len(idxs), "detections": len(keep), "lead_in": True})
                if len(bad):
                    self._finish_library_loop()
                    raise self._index_error(recs["carrier_bin"][stop])
            if not raw.in_lead_in:
                view, first, n = raw.take_rest()
                if n:
                    run(lambda **kw: eng.run_stream(view, first_block_idx=first, **kw), n)
        self._finish_library_loop()
        total = {"blocks": sum(s["blocks"] for s in stats_all),
                 "detections": sum(s["detections"] for s in stats_all), "calls": stats_all}
        recs = None
        if want_records:
            recs = np.concatenate(recs_all) if recs_all else np.zeros(0, dtype=_native.RECORD_DTYPE)
        return total, recs

    def _finish_library_loop(self):
        """Nothing more is read: the window's last locked pages are let go in the background (the
        engine keeps the mapping alive until it is closed), the output is complete NOW."""
        self._exhausted = True
        if self._pin:
            self._engine.input_window_release()
            self._pin = False

    def write_toad(self, output_file):
        """Write the `.toad` lines of every remaining detection to `output_file` (what `thrifty
        detect --quiet -o` does, reference detect.py:217-219).  For a mapped `.card` / raw file the
        whole loop runs inside the library (thr_run_card / thr_run_stream: no Python per batch,
        text written straight to the file's descriptor); any other source takes iter_toad_text().
        -> statistics of the library loop, or None."""
        fd = None
        if self.blocks is not None and self._library_loop_ready():
            try:
                output_file.flush()
                fd = output_file.fileno()
            except (AttributeError, OSError, ValueError):
                fd = None
        if fd is not None:
            stats, _ = self._run_library_loop(out_fd=fd)
            return stats
        binary = "b" in getattr(output_file, "mode", "") or not hasattr(output_file, "encoding")
        for text in self.iter_toad_text():
            output_file.write(text if binary else text.decode("ascii"))
        output_file.flush()
        return None

    def detected_records(self):
        """Every remaining DETECTED record, in input order, each with its timestamp's bits in
        `reserved` -- what the ranks of a sharded run send to rank 0 (parallel.run_sharded).  A
        mapped file runs inside the library like write_toad()."""
        if self.blocks is not None and self._library_loop_ready():
            return self._run_library_loop(want_records=True)[1]
        chunks = []
        for stamps, recs in self.iter_detected_records():
            recs = recs.copy()
            recs["reserved"] = np.ascontiguousarray(stamps, dtype=np.float64).view(np.uint64)
            chunks.append(recs)
        return np.concatenate(chunks) if chunks else np.zeros(0, dtype=_native.RECORD_DTYPE)

    def iter_toad_text(self):
        """Batches of `.toad` text (bytes: '\\n'-terminated lines) for the detected blocks,
        formatted by the engine library (`thr_format_toad`: the text of
        `DetectionResult.serialize()`, byte for byte) -- what `detector_cli --quiet -o` writes."""
        for stamps, recs in self.iter_detected_records():
            yield _native.format_toad(recs, stamps, self.new_len, rxid=self.rxid, with_txid=self._multi,
                                      carrier_offset_f32=_offset_mode(self._offset_type))

    def iter_toad_lines(self):
        """The same as lists of lines (str, no line ends)."""
        for text in self.iter_toad_text():
            yield text.decode("ascii").split("\n")[:-1]

    def next(self):
        """Result for the next block of the `blocks` iterator."""
        if self.blocks is None and not self._ready:
            raise TypeError("Detector was constructed without a block source")
        while not self._ready and self._more():
            self._refill()      # (a batch may contribute nothing under only_detections)
        if not self._ready:
            raise StopIteration
        item = self._ready.popleft()
        if isinstance(item, _Deferred):
            self._ready.clear()
            self._exhausted = True      # the reference's loop died here
            self._drop_ahead()
            raise item.exc
        return item

    def __call__(self, timestamp, block_idx, block):
        self.detect(timestamp, block_idx, block)

    def close(self):
        """Release the engine handle (device buffers, streams, the input window) now instead of at
        garbage collection -- the detector's stage objects refer back to it, so plain reference
        counting does not free it.  Also the exit of `with Detector(...) as det:`."""
        self._drop_ahead()
        self._exhausted = True
        self._engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __iter__(self):
        """Iterating hands out what next() does, a batch's results straight from the batch (a generator
        over `_ready`: one interpreter call per BATCH instead of two per block -- `for detected, result
        in Detector(...)` is the reference's operator loop, detect.py:217); next(detector) between two
        steps of the loop takes from the same queue, in order."""
        if self.blocks is None and not self._ready:
            raise TypeError("Detector was constructed without a block source")
        return self._iterate()

    def _iterate(self):
        ready = self._ready
        while True:
            while ready:
                item = ready.popleft()
                if isinstance(item, _Deferred):
                    ready.appendleft(item)
                    self.next()             # raises it, with next()'s bookkeeping
                yield item
            if not self._more():
                return
            self._refill()

    def __next__(self):
        return self.next()


class MultiTemplateDetector(Detector):
    """Several TX templates correlated per block with ONE carrier stage / FFT#2 (BASELINE
    configs[4]): `settings.template` is [n_templates, template_len].  The reference correlates
    one template (detect.py:40-58); its data model already carries the transmitter as
    `DetectionResult.txid` (toads_data.py:22-61), which is what identifies a template here.

    Iterating yields, per input block, a list of (detected, DetectionResult) -- one per
    template in template order, `txid` = template index (under `only_detections`: the detected
    ones of a block, blocks without any skipped).  `iter_detected_records()` /
    `iter_toad_text()` / `iter_toad_lines()` hand out the detections flat, ordered
    [block][template], with the txid column after the rxid (the `.toads` line layout)."""

    _multi = True

    def __init__(self, settings, blocks=None, rxid=-1, yield_data=False, batch_size=None,
                 device_id=0, max_wait=None, populate_threads=0, low_cpu=False, strict_fit=False):
        if yield_data:
            raise TypeError("stage dumps (yield_data) are a single-template facility")
        super(MultiTemplateDetector, self).__init__(settings, blocks, rxid=rxid, batch_size=batch_size,
                                                    device_id=device_id, max_wait=max_wait,
                                                    populate_threads=populate_threads, low_cpu=low_cpu,
                                                    strict_fit=strict_fit)
        self.n_templates = int(np.asarray(settings.template).shape[0])

    def _flat(self, stamps, idxs, recs):
        t = recs.shape[1]
        return (np.repeat(np.asarray(stamps, dtype=np.float64), t).tolist(), np.repeat(idxs, t),
                recs.reshape(-1))

    # (`txid` = the record's template index: set by the result builder, csrc/fastresults.c)

    def _package(self, results, groups):
        """Flat [block][template] results -> one list per block (`groups`: the flat positions
        that survived an only_detections filter; position // n_templates is the block)."""
        t = self.n_templates
        pos = np.arange(len(results)) if groups is None else np.asarray(groups)
        out, last = [], None
        for item, k in zip(results, (pos // t).tolist()):
            if isinstance(item, _Deferred):
                out.append(item)
                last = None
            elif k == last:
                out[-1].append(item)
            else:
                out.append([item])
                last = k
        return out

    def detect_batch(self, items):
        """[(timestamp, block_idx, block), ...] -> per block a list of (detected, result) per template."""
        if not items:
            return []
        arr = self._stack([it[2] for it in items])
        idx = np.array([int(it[1]) for it in items], dtype=np.int64)
        stamps, idxs, recs = self._flat([it[0] for it in items], idx, self._run(arr, idx))
        flat = self._results(stamps, idxs, recs)
        if flat and isinstance(flat[-1], _Deferred):
            raise flat[-1].exc
        return self._package(flat, None)

    def detect(self, timestamp, block_idx, block):
        """One block -> [(detected, DetectionResult), ...], one per template."""
        return self.detect_batch([(timestamp, block_idx, block)])[0]


def _write_fd(fd, data):
    import os
    view = memoryview(data)
    while len(view):
        view = view[os.write(fd, view):]


def detector_cli(detector_class, parser=None, extra_args=None, argv=None):
    """`thrifty detect` front end (reference detect.py:161-223): same arguments and settings keys;
    `detector_class(settings, blocks, rxid=..., **kwargs)` must iterate to `(detected, result)` pairs.
    The implementation is thrifty_amd/detect_cli.py (which also holds `SummaryLineFormatter`)."""
    from thrifty_amd import detect_cli
    return detect_cli.detector_cli(detector_class, parser, extra_args, argv)


def __getattr__(name):
    if name == "SummaryLineFormatter":      # (reference detect.py:103-158; lives with the CLI)
        from thrifty_amd import detect_cli
        return detect_cli.SummaryLineFormatter
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def _main():
    # (`python -m thrifty_amd.detect` runs this file as __main__: hand the front end the class of the
    # IMPORTED module, the one it compares detector classes with)
    from thrifty_amd import detect, detect_cli
    detect_cli.detector_cli(detect.Detector)


if __name__ == "__main__":
    _main()
