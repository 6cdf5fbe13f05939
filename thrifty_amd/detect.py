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

import argparse
import logging
import sys
import time
from collections import deque, namedtuple
from types import SimpleNamespace

import numpy as np

from thrifty_amd import _native, toads_data, util
try:
    from thrifty_amd import _fastresults
except ImportError as _exc:      # pragma: no cover -- a tree that was never built
    raise ImportError("thrifty_amd._fastresults is not built: run `python -m thrifty_amd.build` (%s)" % _exc)
from thrifty_amd.block_data import CardStream, RawStream, block_reader, card_reader
from thrifty_amd.setting_parsers import normalize_freq_range
from thrifty_amd.settings import load_args

DetectorSettings = namedtuple("DetectorSettings", [
    "block_len", "history_len", "carrier_len", "carrier_thresh", "carrier_window",
    "template", "corr_thresh"])


_LOG = logging.getLogger("thrifty_amd.detect")

_SLOW_SOURCE_S = 0.002   # inter-arrival time above which a LIVE source ends the batch being filled
_UNKNOWN_FILL_S = 0.05   # a source that does not say whether it is live: longest time spent FILLING one batch
_YIELD_DATA_BATCH = 64    # blocks per batch with yield_data (each drags two N-point dumps along)


def unique_window(block_len, history_len, template_len):
    """Half-open range of correlation lags owned by one block (reference
    soa_estimator.py:20-39)."""
    assert history_len >= template_len - 1
    corr_len = block_len - template_len + 1
    pad = history_len - template_len + 1
    return pad // 2, corr_len - (pad - pad // 2)


def _offset_mode(offset_type):
    """thr_format_toad's carrier_offset_f32 argument for a Detector's `_offset_type`."""
    return 0 if offset_type is float else 2 if offset_type is int else 1


class _Deferred(object):
    """An exception that belongs to one position of a batch: raised when that position is
    reached, after the results before it have been delivered."""

    def __init__(self, exc):
        self.exc = exc


class _SyncStage(object):
    """`Detector.sync`: what the reference's `DefaultSynchronizer` offers an analysis script
    (carrier_sync.py:30-118) -- the attributes `thresh_coeffs`, `window`, `weights` and the call
    `sync(signal) -> (shifted_fft or None, CarrierSyncInfo)` -- evaluated by the engine for ONE
    block (carrier stage, Dirichlet fit, frequency shift, FFT#2; `thr_debug_stage` returns the
    shifted spectrum in natural order).  `detector` and `shifter` are stages of fused kernels and
    cannot be replaced (the reference's own subclasses that do so are separate detectors here:
    `PreshiftDetector`, `FastDetector`).  `interpolator` CAN be assigned, as the reference's
    InterpolationDetector does (experimental/detect_carrier_interpol.py:17-40): any callable
    `(fft_mag, peak_idx) -> offset` -- or None for no sub-bin estimate, carrier_sync.py:66-68 --
    then runs on the HOST between two engine passes (carrier stage + |FFT#1| out, offsets back in:
    thr_detect_offsets), a slow path for analysis scripts."""

    _DEVICE_FIT = object()      # `interpolator` not assigned: the engine's own Dirichlet fit (k_fit)

    def __init__(self, det, settings):
        self._det, self.weights = det, None
        self.thresh_coeffs, self.window = settings.carrier_thresh, settings.carrier_window
        # _last: (the shifted_fft handed out, record, corr) of the latest block
        self._last, self._interpolator = None, self._DEVICE_FIT

    @property
    def interpolator(self):
        return self._device_interpolator if self._interpolator is self._DEVICE_FIT else self._interpolator

    @interpolator.setter
    def interpolator(self, fn):
        if fn is not None and not callable(fn):
            raise TypeError("sync.interpolator takes a callable (fft_mag, peak_idx) -> offset, or None")
        self._interpolator = fn
        self._det._use_host_interpolator()

    def sync(self, signal):
        det = self._det
        if self._interpolator is not self._DEVICE_FIT:
            # (the stage dump is the engine's OWN pipeline, Dirichlet fit included: it cannot show
            # the spectrum shifted by somebody else's offset)
            raise NotImplementedError("sync(block) evaluates the engine's own stages; with a replaced "
                                      "interpolator use Detector.detect(timestamp, block_idx, block)")
        arr = det._stack([signal])
        rec = det._run(arr, np.zeros(1, dtype=np.int64))[0, 0]
        _, result = det._result(0.0, 0, rec)
        if result.corr_info is None:
            self._last = None
            return None, result.carrier_info
        xhat, corr = det._engine.debug_stage(arr)
        shifted_fft = xhat[0]
        self._last = (shifted_fft, rec, corr[0][:det.soa_estimate.corr_len])
        return shifted_fft, result.carrier_info

    __call__ = sync

    def detect(self, fft_mag):
        raise NotImplementedError(
            "the carrier detector runs inside the engine's carrier kernel, on a block's samples: "
            "call sync(block) -- or Detector.detect(timestamp, block_idx, block) -- instead")

    detector = detect

    def _device_interpolator(self, fft_mag, peak_idx):
        raise NotImplementedError("the Dirichlet fit runs inside the engine (k_fit): call sync(block) -- or "
                                  "assign sync.interpolator a host callable (slow path)")

    def shifter(self, signal, shift):
        raise NotImplementedError("the frequency shift is fused into the correlate kernel: call sync(block)")


class _SoaStage(object):
    """`Detector.soa_estimate`: the attributes of the reference's `SoaEstimator`
    (soa_estimator.py:63-92: `template`, `template_energy`, `corr_len`, `window`,
    `thresh_coeffs`) and the call `soa_estimate(fft) -> (detected, CorrDetectionInfo, corr)` for
    the spectrum `Detector.sync(block)` has just returned -- the pair of calls that makes up the
    body of the reference's `Detector.detect` (detect.py:60-78).  Any other spectrum would have to be
    correlated from host memory, which the engine has no entry point for."""

    _DEVICE = object()          # `interpolate` not assigned: the engine's log-parabola (k_finish)

    def __init__(self, det, settings, template, corr_len):
        self._det = det
        self._interpolate = self._DEVICE
        self.last_fft = None        # the shifted spectrum of the block a replaced `interpolate` is looking at
        self.template = template
        self.template_energy = float(np.sum(np.abs(template) ** 2))
        self.corr_len = corr_len
        self.thresh_coeffs = settings.corr_thresh
        self.window = unique_window(settings.block_len, settings.history_len, template.shape[-1])

    @property
    def interpolate(self):
        """The correlation-peak interpolator (reference soa_estimator.py:74: `self.interpolate =
        gaussian_interpolation`).  Assignable like the reference's (experimental/
        detect_xcorr_interpol.py:62): any callable `(corr_mag, peak_idx) -> offset`, evaluated on the
        HOST for the detected blocks of a batch -- a slow path for analysis scripts."""
        return self._device_interpolate if self._interpolate is self._DEVICE else self._interpolate

    @interpolate.setter
    def interpolate(self, fn):
        if not callable(fn):
            raise TypeError("soa_estimate.interpolate takes a callable (corr_mag, peak_idx) -> offset")
        self._det._use_host_soa_interpolator()
        self._interpolate = fn

    def _device_interpolate(self, corr_mag, peak_idx):
        raise NotImplementedError("the log-parabola runs inside the engine (k_finish): call "
                                  "soa_estimate(shifted_fft) -- or assign soa_estimate.interpolate a host callable")

    def soa_estimate(self, fft):
        if self._interpolate is not self._DEVICE:
            raise NotImplementedError("soa_estimate(fft) evaluates the engine's own stages; with a replaced "
                                      "interpolator use Detector.detect(timestamp, block_idx, block)")
        last = self._det.sync._last
        if last is None or fft is not last[0]:
            raise NotImplementedError(
                "soa_estimate() takes the shifted spectrum that Detector.sync(block) returned for "
                "the latest block; arbitrary spectra cannot be handed to the engine")
        _, rec, corr = last
        detected = bool(int(rec["flags"]) & _native.FLAG_CORR)
        info = toads_data.CorrDetectionInfo(int(rec["corr_sample"]), float(rec["corr_offset"]) if detected else 0,
                                            float(rec["corr_energy"]), float(rec["corr_noise"]))
        return detected, info, corr

    __call__ = soa_estimate


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
        self.sync = _SyncStage(self, settings)
        self.soa_estimate = _SoaStage(self, settings, template, corr_len)

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


def _carrier_freq(carrier_info, block_len, sample_rate):
    bin_freq = sample_rate / block_len
    return (util.fft_bin(carrier_info.bin, block_len) + carrier_info.offset) * bin_freq


class SummaryLineFormatter(object):
    """One human-readable line per block (reference detect.py:103-158)."""

    def __init__(self, sample_rate, block_len, add_dt=False):
        self.sample_rate = sample_rate
        self.block_len = block_len
        self.add_dt = add_dt

    _CARRIER = ("blk={blk}; carrier: {det} @ {freq:.3f} kHz / {idx:>3.0f}:{offset:+.2f}, "
                "SNR = {ampl:>4.0f} / {noise:>2.0f} = {snr:>5.2f} dB")
    _CORR = "; corr: {det} @ {idx:>4}{offset:+.3f}{dt}, SNR = {ampl:>4.0f}/{noise:>2.0f} = {snr:>5.2f} dB"

    @staticmethod
    def _stage(fmt, mark, info, **extra):
        """One stage's half of the line: its verdict mark, its four-field info tuple and its SNR."""
        return fmt.format(det=mark, idx=info[0], offset=info[1], ampl=info[2], noise=info[3],
                          snr=util.snr(info[2], info[3]), **extra)

    def __call__(self, detected, result):
        has_carrier = result.corr_info is not None
        parts = [self._stage(self._CARRIER, "yes" if has_carrier else "no ", result.carrier_info, blk=result.block,
                             freq=_carrier_freq(result.carrier_info, self.block_len, self.sample_rate) / 1e3)]
        if has_carrier:
            parts.append(self._stage(self._CORR, "yes" if detected else "no ", result.corr_info, dt=""))
        return "".join(parts)


def _strip_output_args(argv):
    """argv without -o/--output/-a/--append (ranks other than 0 must not open -- and with -o
    truncate -- the output file that rank 0 writes)."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("-o", "--output", "-a", "--append"):
            skip = True
            continue
        if a.startswith(("--output=", "--append=")) or (a[:2] in ("-o", "-a") and len(a) > 2 and a[1] != "-"):
            continue
        out.append(a)
    return out


def detector_cli(detector_class, parser=None, extra_args=None, argv=None):
    """`thrifty detect` front end (reference detect.py:161-223): same arguments and
    settings keys; `detector_class(settings, blocks, rxid=..., **kwargs)` must iterate
    to `(detected, result)` pairs.

    Addition: `--gpus N` shards a regular input file over N GPUs of this node by contiguous
    block ranges, one process per GPU (re-launched under `torch.distributed.run`); the ranks'
    detection records are gathered to rank 0 over RCCL and written as ONE `.toad` in input
    order.  The per-block summary lines are a console aid of the
    single-process loop and are not printed in that mode."""
    from thrifty_amd import parallel
    parallel.rank_env()     # (before the HIP runtime initialises: the same on every launch route)
    argv = list(sys.argv[1:] if argv is None else argv)
    gpus = parallel.peek_gpus(argv)
    # a rank of a sharded run is a process that THIS CLI re-launched, or that torchrun started
    # with as many ranks as --gpus asks for; RANK / WORLD_SIZE inherited from an unrelated
    # launcher do not turn a plain `thrifty detect` into one (parallel.sharded_env)
    rank, world, local = parallel.sharded_env(gpus)
    if gpus > 1 and world is None:
        if not any(a in ("-o", "--output", "-a", "--append") or a.startswith(("--output=", "--append="))
                   or (a[:2] in ("-o", "-a") and len(a) > 2 and a[1] != "-") for a in argv):
            raise SystemExit("--gpus %d needs an output file (-o / -a): the ranks' detections are "
                             "gathered and written by rank 0, nothing is printed per block" % gpus)
        sys.exit(parallel.relaunch_under_torchrun(gpus, argv))
    if world is not None and rank != 0:
        argv = _strip_output_args(argv)
    if parser is None:
        parser = argparse.ArgumentParser(description=__doc__,
                                         formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("input", type=argparse.FileType("rb"), default="-",
                        help="input data ('-' streams from stdin)")
    parser.add_argument("--raw", dest="raw", action="store_true", help="input data is raw binary data")
    parser.add_argument("--quiet", dest="quiet", action="store_true",
                        help="do not write anything to standard output")
    parser.add_argument("--gpus", dest="gpus", type=int, default=1,
                        help="shard a regular input file over this many GPUs of the node")
    parser.add_argument("--dist-backend", dest="dist_backend", choices=["nccl", "gloo"], default="nccl",
                        help="with --gpus N: nccl = RCCL, one GPU per rank (the real thing); gloo = a "
                             "rehearsal of the N-rank run on ONE GPU (every rank computes on device 0, "
                             "the records travel over gloo)")
    parser.add_argument("--templates", dest="templates", nargs="+", metavar="NPY", default=None,
                        help="correlate every block against several TX templates (.npy files of "
                             "equal length, instead of the `template` setting); detections carry "
                             "the template's position as txid, written after the rxid")
    group = parser.add_mutually_exclusive_group()
    group.add_argument("-o", "--output", dest="output", type=argparse.FileType("w"),
                       help="Output file (.toad) ('-' for stdout)")
    group.add_argument("-a", "--append", dest="append", type=argparse.FileType("a"),
                       help="Output file to append to (.toad)")
    keys = ["sample_rate", "block_size", "block_history", "carrier_window", "carrier_threshold",
            "corr_threshold", "template", "rxid"]
    config, args = load_args(parser, keys, argv=argv)
    kwargs = {a: args[a] for a in extra_args} if extra_args is not None else {}

    output_file = args.output if args.append is None else args.append
    info_out = sys.stderr if output_file is sys.stdout else sys.stdout
    window = normalize_freq_range(config.carrier_window, config.sample_rate / config.block_size)
    if args.raw:
        # byte stream -> overlapping blocks framed on the device (block_reader-compatible
        # tuples if the detector class iterates it the classic way)
        blocks = RawStream(args.input, config.block_size, config.block_history)
    else:
        # binary stream -> batches with on-device base64 decode (card_reader-compatible tuples
        # if the detector class iterates it the classic way)
        blocks = CardStream(args.input, config.block_size)
    if args.templates:
        tpls = [np.load(f) for f in args.templates]
        if len({t.shape for t in tpls}) != 1 or tpls[0].ndim != 1:
            raise SystemExit("--templates: the templates must be 1-D arrays of one length")
        template = np.stack(tpls)
        if detector_class is Detector:
            detector_class = MultiTemplateDetector
    else:
        template = np.load(config.template)
    settings = DetectorSettings(block_len=config.block_size, history_len=config.block_history,
                                carrier_len=template.shape[-1], carrier_thresh=config.carrier_threshold,
                                carrier_window=window, template=template,
                                corr_thresh=config.corr_threshold)
    if world is not None:
        # one rank of a sharded run (also world == 1 under torchrun: same code path, same collectives)
        if args.gpus != world:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        blocks.shard(rank, world)
        if "populate_threads" not in kwargs and detector_class in (Detector, MultiTemplateDetector):
            kwargs["populate_threads"] = parallel.populate_threads(world)   # the ranks share the host's CPUs
            kwargs["low_cpu"] = parallel.cpu_budget() // max(1, world) < 4  # (fewer than 4 CPUs per rank)
        if args.dist_backend == "gloo":
            local = 0               # rehearsal: every rank computes on device 0
        # torch, the process group and the gather rehearsal first: the engine's threads start with it
        parallel.init_rank(rank, world, local, backend=args.dist_backend)
        detections = detector_class(settings, blocks, rxid=config.rxid, device_id=local, **kwargs)
        try:
            if not hasattr(detections, "iter_detected_records"):
                raise SystemExit("--gpus: %s does not expose iter_detected_records() (the records that "
                                 "travel between the ranks)" % type(detections).__name__)
            if getattr(detections, "_host_path", False):
                # (every rank builds the same class, so every rank leaves here)
                raise SystemExit("--gpus: %s replaces a stage by a host callable (sync.interpolator / "
                                 "soa_estimate.interpolate): that slow path runs in one process, not sharded"
                                 % type(detections).__name__)
            parallel.run_sharded(detections, rank, world, local, output_file, backend=args.dist_backend)
        finally:
            _close(detections)
        return
    detections = detector_class(settings, blocks, rxid=config.rxid, **kwargs)
    try:
        _cli_loop(detections, args, config, output_file, info_out)
    finally:
        # the engine goes NOW (threads joined, pages unlocked, device memory back), not whenever the
        # interpreter's shutdown gets to an object whose stages refer back to it
        _close(detections)


def _close(detections):
    close = getattr(detections, "close", None)
    if callable(close):
        close()


def _cli_loop(detections, args, config, output_file, info_out):
    """The reference's loop (detect.py:214-223) over whatever detector class the caller passed."""
    if args.quiet and hasattr(detections, "only_detections"):
        detections.only_detections = True   # nothing is printed for the other blocks anyway
    if (args.quiet and output_file is not None and hasattr(detections, "write_toad")
            and not getattr(detections, "_host_path", False)):
        # nothing per block is needed: a mapped input runs entirely inside the library
        # (thr_run_card / thr_run_stream), anything else a batch of text at a time
        detections.write_toad(output_file)
        return
    if (args.quiet and output_file is not None and hasattr(detections, "iter_toad_text")
            and not getattr(detections, "_host_path", False)):
        for text in detections.iter_toad_text():
            output_file.write(text.decode("ascii"))
        output_file.flush()
        return
    summary = SummaryLineFormatter(config.sample_rate, config.block_size, add_dt=True)
    for item in detections:
        # (one (detected, result) pair per block; a multi-template detector: a list of them)
        for detected, result in (item if isinstance(item, list) else [item]):
            if detected and output_file is not None:
                print(result.serialize(), file=output_file)
            if not args.quiet:
                line = summary(detected, result)
                print(line if result.txid is None else "tx=%d; %s" % (result.txid, line), file=info_out)
    if output_file is not None:
        output_file.flush()


def _main():
    detector_cli(Detector)


if __name__ == "__main__":
    _main()
