"""Randomised configurations inside the suite (the full tool is tests/tools/fuzz_parity.py):
random block length (512 ... 65536: the short-block, 16384, long-block and multi-pass kernels),
history, template kind and length, carrier window (negative / wrapping / full), thresholds with
and without stddev terms, batch split, u8 or complex64 input -- five blocks each, GPU vs oracle.
Templates are kept at W >= N / 24 so that the carrier fit is well conditioned."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_random_configurations_equal_the_oracle(seed, monkeypatch):
    import fuzz_parity
    monkeypatch.setenv("FUZZ_MIN_RATIO", "24")
    rng = np.random.default_rng(seed)
    failures, ran = [], 0
    for k in range(40):
        status, desc = fuzz_parity.one(rng, k)
        if status == "refused-both":
            continue
        ran += 1
        if status != "ok":
            failures.append((k, desc, status))
    assert ran >= 30
    assert not failures, failures
