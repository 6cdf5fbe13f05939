"""The carrier fit on the device is a port of MINPACK lmdif (csrc/lmdif8.hpp).  Pin the
algorithm: the scalar restatement the port follows (oracle/minpack_lmdif.py) must reproduce
scipy.optimize.curve_fit -- what the reference calls (carrier_sync.py:189) -- on the
reference's own fit problem, including ill-conditioned ones (short templates)."""
import numpy as np
import pytest
from scipy.optimize import curve_fit

from oracle import thrifty_np as onp
from oracle.minpack_lmdif import lmdif


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_restatement_equals_curve_fit(seed):
    rng = np.random.default_rng(seed)
    xd = np.arange(-3, 4)
    checked = 0
    for _ in range(150):
        n = int(rng.choice([1024, 4096, 16384, 65536]))
        w = int(rng.integers(n // 256, n // 2))                 # down to very flat main lobes
        amp, off = rng.uniform(10, 300), rng.uniform(-0.7, 0.7)
        y = (amp * np.abs(onp.dirichlet(xd - off, n, w)) + rng.normal(0, amp * 0.03, 7))
        y = y.astype(np.float32).astype(np.float64)             # magnitudes arrive as float32

        def model(x, a, o):
            return a * np.abs(onp.dirichlet(np.array(x, dtype=np.float64) - o, n, w))
        try:
            popt, _ = curve_fit(model, xd, y, p0=(y[3], 0))
        except RuntimeError:
            continue                                            # maxfev reached in SciPy
        sol, info, nfev = lmdif(lambda p: list(model(xd, p[0], p[1]) - y), [y[3], 0.0], 7)
        assert info in (1, 2, 3, 4)
        assert sol[0] == popt[0] and sol[1] == popt[1]          # to the last bit
        checked += 1
    assert checked > 140
