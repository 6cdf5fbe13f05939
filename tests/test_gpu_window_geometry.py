"""The peak search of k_correlate is specialised on how many 1024-lag rows lie outside / inside
the unique window (csrc/detect16k.hip, RLO / RHI variants).  These cases put the window edges on,
just before and just after row boundaries -- so that each (history, template) pair selects the
(1, 2) variant, the (0, 4) variant or the generic kernel -- and plant bursts ON the window's
first and last lags and next to them; every record must equal the oracle's."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_util  # noqa: E402
from oracle import thrifty_np as onp  # noqa: E402
from thrifty_amd import _native as F  # noqa: E402
from thrifty_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu

N = 16384

# (history, template length) -> unique window [lo, hi) = [pad // 2, N - W + 1 - (pad - pad // 2)), pad = H - W + 1
CASES = [
    (4096, 1023),    # BASELINE: [1537, 13825)            -> rows 0, 14, 15 outside: variant (1, 2)
    (3070, 1023),    # [1024, 14338): row 0's last lag 1023 == lo - 1 (the peak's left neighbour) -> generic
    (3072, 1023),    # [1025, 14337): row 0 outside, rows 1 .. 13 inside, row 14 holds one lag
    (5120, 1023),    # [2049, 13313): rows 0, 1 and 14, 15 outside; row 13 holds lag 13312 only
    (5118, 1023),    # [2048, 13314)
    (4920, 4914),    # example detector.cfg: [3, 11467)   -> rows 12 .. 15 outside: variant (0, 4)
    (5099, 4096),    # [502, 11787)
    (1100, 1023),    # [39, 15323): nothing outside -> generic
    (8200, 1023),    # [3589, 11773): more rows outside than either variant assumes (still valid for (1, 2))
]


@pytest.mark.parametrize("h,w", CASES)
def test_bursts_on_the_window_edges_equal_the_oracle(h, w):
    rng = np.random.default_rng(h * 7 + w)
    tpl = np.sign(rng.normal(0, 1, w))
    lo, hi = onp.unique_window(N, h, w)
    edge = [lo, lo + 1, lo + 2, hi - 1, hi - 2, hi - 3, (lo + hi) // 2,
            ((lo // 1024) + 1) * 1024 - 1, ((lo // 1024) + 1) * 1024, (hi // 1024) * 1024 - 1, (hi // 1024) * 1024]
    edge = [p for p in edge if lo <= p < hi]
    pos = np.array(edge * 3 + list(rng.integers(lo, hi, 40)))
    nb = len(pos)
    blocks, truth = synth.synth_blocks(rng, nb, N, tpl, (lo, hi), signal_frac=1.0, positions=pos,
                                       carrier_bins=(12.0, 100.0))
    thr = (0, 15, 0)
    eng = F.Engine(N, h, tpl, thr, (7, 110), thr, max_batch=64)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    rows = soak_util.run_oracle(blocks, N, h, tpl, thr, (7, 110), thr, procs=8, chunk=16)
    mism, worst, ties = soak_util.compare(rec, rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR)
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst, ties)
    assert not ties
    assert worst["offset"] <= 5e-6 and worst["energy"] <= 2e-5 and worst["noise"] <= 2e-5, worst
    found = rec["corr_sample"][(rec["flags"] & F.FLAG_CORR) != 0]
    assert len(found) >= nb - 2
    assert {int(lo), int(hi - 1)} <= set(found.tolist())      # peaks ON both window edges were found
