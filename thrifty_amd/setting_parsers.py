"""String -> value converters for detector settings.

Grammar follows the reference (thrifty/setting_parsers.py:43-185): floats with an
SI suffix, frequency ranges in bins or Hz ("7 - 110", "50-60 kHz"), and threshold
formulas "a + b*snr + c*stddev" (symbols: constant|c, snr|s, stddev|d).
"""
from __future__ import annotations

import re

_NUM = r"[-+]?(?:\d+(?:\.\d*)?|\.\d+)(?:[eE][-+]?\d+)?"
_RANGE_RE = re.compile(r"^(%s)(?:\s*-\s*(%s))?\s*([kKmM]?)([hH][zZ])?$" % (_NUM, _NUM), re.IGNORECASE)
_TERM_RE = re.compile(r"^\s*(?=\S)(?:(%s)\s*\*?\s*)?(constant|c|snr|s|stddev|d|)\s*$" % _NUM)

SI_PREFIXES = {"y": 1e-24, "z": 1e-21, "a": 1e-18, "f": 1e-15, "p": 1e-12, "n": 1e-9,
               "u": 1e-6, "m": 1e-3, "c": 1e-2, "d": 1e-1, "k": 1e3, "M": 1e6, "G": 1e9,
               "T": 1e12, "P": 1e15, "E": 1e18, "Z": 1e21, "Y": 1e24}


def metric_float(string):
    """'2.4M' -> 2400000.0, '3.4m' -> 0.0034, '123.4' -> 123.4."""
    text = string.strip()
    scale = SI_PREFIXES.get(text[-1:])
    return float(text) if scale is None else float(text[:-1]) * scale


def freq_range(string):
    """-> (start, stop, in_hertz).  A single number means start == stop."""
    m = _RANGE_RE.match(string)
    if m is None:
        raise ValueError("Invalid range: {}".format(string))
    lo, hi, mag, unit = m.groups()
    start = float(lo)
    stop = start if hi is None else float(hi)
    mult = {"k": 1e3, "m": 1e6}.get(mag.lower(), None)
    if mult is not None:
        start, stop = start * mult, stop * mult
    return start, stop, unit is not None


def normalize_freq_range(range_, bin_freq):
    """(start, stop, in_hertz) -> integer (start_bin, stop_bin)."""
    start, stop, in_hertz = range_
    if in_hertz:
        return int(start / bin_freq), int(stop / bin_freq)
    return int(start), int(stop)


def threshold(string):
    """'5 + 3*snr + stddev' -> (5.0, 3.0, 1.0); '10c+5s+2d' -> (10.0, 5.0, 2.0)."""
    if not string:
        raise ValueError("Empty string")
    acc = {"c": 0.0, "s": 0.0, "d": 0.0}
    for term in string.split("+"):
        m = _TERM_RE.match(term)
        if m is None:
            raise ValueError("Invalid threshold term: {}".format(term))
        qty, sym = m.groups()
        key = {"": "c", "constant": "c", "c": "c", "snr": "s", "s": "s", "stddev": "d", "d": "d"}[sym]
        acc[key] += 1.0 if qty is None else float(qty)
    return acc["c"], acc["s"], acc["d"]
