"""The peak search of k_correlate is specialised on how many 1024-lag rows lie outside / inside
the unique window (csrc/correlate16k_geom.hpp: the table RLO = 0 .. 2 x RHI = 0 .. 4, one kernel
each).  These cases put the window edges on, just before and just after row boundaries -- every
(history, template) pair names the table entry it must select, or the generic kernel -- and plant
bursts ON the window's first and last lags and next to them; every record must equal the
oracle's, and the specialised launch must equal the generic kernel's records byte for byte."""
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

# (history, template length, table entry) -> unique window [lo, hi) = [pad // 2, N - W + 1 - (pad - pad // 2)), pad = H - W + 1
CASES = [
    (4096, 1023, (1, 2)),    # BASELINE: [1537, 13825)     -> rows 0, 14, 15 outside
    (3070, 1023, (0, 1)),    # [1024, 14338): row 0's last lag 1023 == lo - 1 (the peak's left neighbour) keeps row 0
    (3072, 1023, (1, 1)),    # [1025, 14337): row 0 outside, rows 1 .. 13 inside, row 14 holds one lag
    (5120, 1023, (2, 2)),    # [2049, 13313): rows 0, 1 and 14, 15 outside; row 13 holds lag 13312 only
    (5118, 1023, (1, 2)),    # [2048, 13314): lag 2047 is the left neighbour of the window's first lag
    (4920, 4914, (0, 4)),    # example detector.cfg: [3, 11467) -> rows 12 .. 15 outside
    (5099, 4096, (0, 4)),    # [502, 11787)
    (1100, 1023, (0, 1)),    # [39, 15323): row 15 outside
    (1022, 1023, (0, 0)),    # [0, 15362): the window is every kept lag; rows 1 .. 14 skip the test
    (2100, 2000, (0, 2)),    # [50, 14334)
    (3200, 3100, (0, 3)),    # [50, 13234)
    (2300, 200, (1, 1)),     # a short template: [1050, 15134)
    (7000, 3000, (1, 4)),    # [2000, 11384)
    (6200, 1200, (2, 3)),    # [2500, 12684)
    (7200, 2000, (2, 4)),    # [2600, 11784)
    (8200, 1023, None),      # [3589, 11773): starts in row 3 -- outside the table, the generic kernel
    (8190, 3000, None),      # [2595, 10789): ends in row 10
]


def test_the_table_covers_every_window_it_claims():
    """Host-side restatement of the selection rule: for EVERY (history, template) pair, the entry
    (lo, hi) is the one with w_lo in row lo and w_hi in row 15 - hi (one lag of margin for the
    peak's neighbours) -- checked against the engine's own answer on a spread of geometries."""
    rng = np.random.default_rng(5)
    seen = set()
    for _ in range(60):
        w = int(rng.integers(16, 6000))
        h = int(rng.integers(w - 1, min(N - 1, w + 9000)))
        lo, hi = onp.unique_window(N, h, w)
        if hi <= lo:
            continue
        rows_lo = (lo - 1) // 1024 if lo > 0 else 0          # rows whose last lag is below lo - 1
        rows_hi = (N - hi - 1) // 1024 if hi < N else 0      # rows whose first lag is above hi
        want = (rows_lo, rows_hi) if rows_lo <= 2 and rows_hi <= 4 else (-1, -1)
        # (short templates run sectioned by default -- detect16k_sec.hip; the table is k_correlate's)
        eng = F.Engine(N, h, np.sign(rng.normal(0, 1, w)), (0, 15, 0), (7, 110), (0, 15, 0), max_batch=1,
                       path="unsectioned")
        assert eng.correlate_geom() == want, (h, w, lo, hi)
        eng.close()
        seen.add(want)
    assert len(seen) >= 8


@pytest.mark.parametrize("h,w,geom", CASES)
def test_bursts_on_the_window_edges_equal_the_oracle(h, w, geom):
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
    # (k_correlate itself: a one-template handle of a short template would run sectioned by default,
    # tests/test_gpu_sections16k.py)
    eng = F.Engine(N, h, tpl, thr, (7, 110), thr, max_batch=64, path="unsectioned")
    assert eng.correlate_geom() == (geom or (-1, -1)) and eng.sections() == (0, 0)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    # the same blocks through the generic kernel (window test in all 16 rows): equal byte for byte;
    # so are four templates per block (the MULTI variants of the same table entry) and complex64 input
    gen = F.Engine(N, h, tpl, thr, (7, 110), thr, max_batch=64, path="unsectioned_generic_rows")
    assert gen.correlate_geom() == (-1, -1)
    assert gen.detect(blocks, np.arange(nb))[:, 0].tobytes() == rec.tobytes()
    tpl4 = np.stack([tpl, -tpl[::-1], np.roll(tpl, 7), tpl * np.sign(rng.normal(0, 1, w))])
    c64 = ((blocks[:24].astype(np.float32) - 127.4) / 128).view(np.complex64)
    for data in (blocks[:24], c64):
        a = F.Engine(N, h, tpl4, thr, (7, 110), thr, max_batch=64)
        b = F.Engine(N, h, tpl4, thr, (7, 110), thr, max_batch=64, path="generic_rows")
        assert a.correlate_geom() == (geom or (-1, -1)) and b.correlate_geom() == (-1, -1)
        ra, rb = a.detect(data, np.arange(24)), b.detect(data, np.arange(24))
        assert ra.tobytes() == rb.tobytes() and ra.shape == (24, 4)
        if data is blocks:
            assert ra[:, 0].tobytes() == rec[:24].tobytes()
    rows = soak_util.run_oracle(blocks, N, h, tpl, thr, (7, 110), thr, procs=8, chunk=16)
    mism, worst, ties = soak_util.compare(rec, rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR)
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst, ties)
    assert not ties
    assert worst["offset"] <= 5e-6 and worst["energy"] <= 2e-5 and worst["noise"] <= 2e-5, worst
    found = rec["corr_sample"][(rec["flags"] & F.FLAG_CORR) != 0]
    assert len(found) >= nb - 2
    assert {int(lo), int(hi - 1)} <= set(found.tolist())      # peaks ON both window edges were found
