"""Synthetic positioning-signal workloads (templates + IQ blocks).

Used by bench.py and the tests to build the SURVEY.md section 8(d) inputs:
Gold-code templates (same code family as reference gold.py:15-82 /
template_generate.py:39-45, re-derived here as a Fibonacci LFSR pair) and
OOK-modulated, carrier-offset, AWGN-corrupted, u8-quantised IQ blocks
(modulation as reference tests/test_soa_estimator.py:18; quantiser as
block_data.py:55-67).
"""
from __future__ import annotations

import numpy as np

# Preferred-pair feedback taps per register length (exponents other than 0, n).
_PREFERRED = {
    5: ((2,), (1, 2, 3)),
    6: ((5,), (1, 4, 5)),
    7: ((4,), (4, 5, 6)),
    8: ((1, 2, 3, 6, 7), (1, 2, 7)),
    9: ((5,), (3, 5, 6)),
    10: ((2, 5, 9), (3, 4, 6, 8, 9)),
    11: ((9,), (3, 6, 9)),
}


def _msequence(nbits, taps):
    """Maximal-length sequence of a Fibonacci LFSR seeded with all ones: the register is an integer
    whose bit j is chip i + j; the chip entering at the top is the parity of the tapped bits (bit 0 and
    the feedback exponents)."""
    mask = sum(1 << t for t in (0,) + tuple(taps))
    register, chips = (1 << nbits) - 1, []
    for _ in range((1 << nbits) - 1):
        chips.append(register & 1)
        register = (register >> 1) | ((bin(register & mask).count("1") & 1) << (nbits - 1))
    return np.array(chips, dtype=bool)


def gold_code(nbits, index):
    """index-th Gold code (boolean chips) of length 2**nbits - 1: the two m-sequences of the preferred
    pair themselves (0, 1), then the first combined with the second rotated by index - 2 chips."""
    pair = _PREFERRED.get(nbits)
    if pair is None:
        raise ValueError("Preferred pairs for %d bits unknown." % nbits)
    first, second = (_msequence(nbits, taps) for taps in pair)
    return {0: first, 1: second}.get(index, first ^ np.roll(second, 2 - index))


def gold_template(nbits, index, sps=1.0):
    """+-1 template sampled at `sps` samples per chip (integer sampler)."""
    code = gold_code(nbits, index)
    n = int(sps * len(code))
    pick = np.arange(n) * len(code) // n
    return np.where(code, 1, -1)[pick]


def quantise_iq(z):
    """complex -> interleaved u8 I/Q, scale 128, offset 127.4, truncating."""
    f = np.asarray(z).astype(np.complex64).view(np.float32) * 128 + 127.4
    return f.astype(np.uint8)


def synth_blocks(rng, n_blocks, block_len, template, window, *, signal_frac=1.0,
                 amp=0.3, sigma=0.02, carrier_bins=(10.0, 100.0), positions=None,
                 carriers=None):
    """Return (u8[n_blocks, 2*block_len], truth dict).

    Each signal-bearing block holds one OOK burst ``amp*(t+1)/2`` of the
    template at a lag drawn uniformly from the unique window [lo, hi), mixed to
    a fractional carrier bin drawn from `carrier_bins`, plus complex AWGN of
    `sigma` per component, quantised to u8.
    """
    template = np.asarray(template, dtype=np.float64)
    w = len(template)
    lo, hi = window
    has = rng.random(n_blocks) < signal_frac
    pos = rng.integers(lo, hi, n_blocks) if positions is None else np.asarray(positions)
    car = (rng.uniform(carrier_bins[0], carrier_bins[1], n_blocks)
           if carriers is None else np.asarray(carriers, dtype=np.float64))
    out = np.empty((n_blocks, 2 * block_len), dtype=np.uint8)
    ook = amp * (template + 1) / 2
    n = np.arange(w)
    for b in range(n_blocks):
        z = (rng.normal(0, sigma, block_len) + 1j * rng.normal(0, sigma, block_len))
        if has[b]:
            p = int(pos[b])
            z[p:p + w] += ook * np.exp(2j * np.pi * car[b] * (n + p) / block_len)
        out[b] = quantise_iq(z)
    return out, {"has_signal": has, "position": pos, "carrier": car}
