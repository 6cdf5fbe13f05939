"""Small helpers used by the summary line (reference thrifty/util.py:6-22)."""
import numpy as np


def snr(peak_ampl, noise_rms):
    """Amplitude ratio in dB."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return 20 * np.log10(np.divide(peak_ampl, noise_rms))


def fft_bin(idx, fft_len):
    """Index into a standard-order FFT -> signed frequency bin."""
    if idx < 0 or idx <= (2 * fft_len - 1) / 4:
        return idx
    return idx - fft_len
