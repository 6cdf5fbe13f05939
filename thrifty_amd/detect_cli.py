"""The `thrifty detect` front end (reference detect.py:94-223): per-block summary lines, the argument
parser, the single-process loop and the `--gpus N` launch.  `thrifty_amd.detect.detector_cli` is the
reference's name for `detector_cli` here."""
from __future__ import annotations

import argparse
import sys

import numpy as np

from thrifty_amd import util
from thrifty_amd.block_data import CardStream, RawStream
from thrifty_amd.detect import Detector, DetectorSettings, MultiTemplateDetector
from thrifty_amd.setting_parsers import normalize_freq_range
from thrifty_amd.settings import load_args


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


_DETECT_DOC = "Detect positioning signals and estimate sample-of-arrival -- on an MI355X (thrifty_amd.detect)."


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
        parser = argparse.ArgumentParser(description=_DETECT_DOC,
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
