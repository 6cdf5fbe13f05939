"""Detector settings: `key: value` config file + argparse overrides.

Python-3 counterpart of the part of reference thrifty/settings.py the detect
path uses (DEFINITIONS 23-109, load 170-231, load_args 234-306,
parse_kvconfig 309-321): same keys, flags, defaults and error types.
"""
from __future__ import annotations

import logging
from collections import namedtuple

from thrifty_amd import setting_parsers as sp

Definition = namedtuple("SettingDefinition", "args parser default description")

# key -> (long flag, short flag, value parser, default as the user would type it, help).
# Keys, flags and defaults are the reference's CLI/config contract; the help text is ours.
_TABLE = (
    ("sample_rate", "--sample-rate", "-s", sp.metric_float, "2.4M",
     "ADC rate of the receiver in samples per second; SI suffixes allowed"),
    ("chip_rate", "--chip-rate", "-p", sp.metric_float, "0.999707M",
     "chips per second of the transmitted spreading code"),
    ("tuner_freq", "--freq", "-f", sp.metric_float, "433.83M",
     "frequency the SDR tuner is centred on, in Hz"),
    ("tuner_gain", "--gain", "-g", float, "0", "SDR tuner gain in dB"),
    ("capture_skip", "--skip", "-k", int, "1",
     "how many blocks the capture tool discards before it starts writing"),
    ("block_size", "--block-size", "-b", int, "16384",
     "samples per processing block; a power of two"),
    ("block_history", "--history", "-y", int, "4920",
     "samples of overlap: the tail of every block is replayed at the head of the next"),
    ("carrier_window", "--carrier-window", "-w", sp.freq_range, "0--1",
     "where to search for the carrier, as FFT bins or as frequencies (e.g. '7-110', '-20k - 20k')"),
    ("carrier_threshold", "--carrier-threshold", "-t", sp.threshold, "15*snr",
     "detection rule for the carrier peak: constant + k*snr + k*stddev"),
    ("corr_threshold", "--corr-threshold", "-u", sp.threshold, "15*snr",
     "detection rule for the correlation peak, same grammar"),
    ("template", "--template", "-z", str, "template.npy",
     "path of the .npy file holding the sampled positioning code"),
    ("rxid", "--rxid", "-r", int, -1, "integer naming this receiver in the output"),
)
DEFINITIONS = {key: Definition([long_flag, short_flag], parser, default, text)
               for key, long_flag, short_flag, parser, default, text in _TABLE}

DEFAULT_CONFIG_PATH = "detector.cfg"
CONFIG_COMMENT_CHAR = "#"
CONFIG_DELIMITER = ":"
CONFIG_DEST = "config"


class Error(Exception):
    """Base class for settings errors."""


class ConfigSyntaxError(Error):
    def __init__(self, line_no, msg):
        Error.__init__(self)
        self.line_no, self.msg = line_no, msg

    def __str__(self):
        return "line #%d: %s" % (self.line_no, self.msg)


class SettingKeyError(Error):
    def __init__(self, msg):
        Error.__init__(self)
        self.msg = msg

    def __str__(self):
        return repr(self.msg)


class Namespace(dict):
    """dict whose items are also attributes."""

    def __init__(self, mapping):
        dict.__init__(self, mapping)
        self.__dict__.update(mapping)


def parse_kvconfig(config_file):
    """`key: value` lines -> dict; `#` starts a comment, blank lines are skipped, a line without the
    delimiter is a ConfigSyntaxError naming it."""
    def fields(numbered):
        line_no, raw = numbered
        text = (raw.decode() if isinstance(raw, bytes) else raw).partition(CONFIG_COMMENT_CHAR)[0]
        key, delimiter, value = text.partition(CONFIG_DELIMITER)
        if text.strip() and not delimiter:
            raise ConfigSyntaxError(line_no, "No delimiter found")
        return (key.strip(), value.strip()) if delimiter else None

    return dict(filter(None, map(fields, enumerate(config_file, 1))))


def _table(definitions):
    return DEFINITIONS if definitions is None else definitions


def _require_known(keys, table, what):
    for key in keys:
        if key not in table:
            raise SettingKeyError("Unknown {}: {}".format(what, key))


def add_argparse_arguments(parser, keys, definitions=None):
    table = _table(definitions)
    _require_known(keys, table, "key")
    for key, d in ((k, table[k]) for k in keys if table[k].args):
        suffix = "" if d.default is None else " [default: {}]".format(d.default)
        parser.add_argument(*d.args, dest=key, type=str, help=str(d.description) + suffix)


def load(args=None, config_file=None, definitions=None):
    """defaults <- config file <- explicit args; every value parsed by its definition."""
    table = _table(definitions)
    layers = [{k: d.default for k, d in table.items() if d.default is not None},
              parse_kvconfig(config_file) if config_file is not None else {},
              args or {}]
    for layer in layers[1:]:
        _require_known(layer, table, "setting")
    merged = {k: v for layer in layers for k, v in layer.items()}
    return {k: (table[k].parser(v) if isinstance(v, str) else v) for k, v in merged.items()}


def _open_config(path):
    """The config file of a run: an explicit -c PATH, else ./detector.cfg if there is one, else None."""
    if path is not None:
        logging.info("Loaded config file from %s", path)
        return open(path)
    try:
        handle = open(DEFAULT_CONFIG_PATH)
    except IOError:
        logging.warning("No config file found. Using default values.")
        return None
    logging.info("Loaded default config file from %s", DEFAULT_CONFIG_PATH)
    return handle


def load_args(parser, keys, argv=None, definitions=None):
    """Add -v/-c and the settings' own flags to `parser`, parse, load the config
    (explicit -c, else ./detector.cfg if present), return (settings, extra_args)."""
    table = _table(definitions)
    parser.add_argument("-v", "--verbose", help="Increase output verbosity", action="store_true")
    parser.add_argument("-c", "--config", dest=CONFIG_DEST, type=str, default=None,
                        help="Config file to load settings from [default: {}]".format(DEFAULT_CONFIG_PATH))
    add_argparse_arguments(parser, keys, definitions=table)
    parsed = vars(parser.parse_args(argv))
    logging.basicConfig(level=logging.DEBUG) if parsed["verbose"] else None
    config_file = _open_config(parsed.pop(CONFIG_DEST))
    try:
        values = load({k: parsed[k] for k in keys if parsed.get(k) is not None}, config_file, table)
    finally:
        config_file.close() if config_file is not None else None
    own = set(keys)
    return (Namespace({k: v for k, v in values.items() if k in own}),
            Namespace({k: v for k, v in parsed.items() if k not in own}))
