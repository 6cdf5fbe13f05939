"""block_len 16384 with one short template: THR_PATH_AUTO runs the correlate stage as up to four
overlap-save sections of 4096 samples (csrc/detect16k_sec.hip).  Every record must equal the
oracle's, and the unsectioned kernel's (path="unsectioned": k_correlate) in every exact field and to
float32 rounding in the others -- on bursts ON the window's edges and on both sides of every
section boundary, for geometries with four, three and partly owned sections, for u8 and complex64
input, packed blocks and raw-stream framing; the whole-rows peak search must equal the generic one
(path="generic_rows") byte for byte."""
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
THR = (0, 15, 0)

# (history, template length, sections)
CASES = [
    (4096, 1023, 4),     # BASELINE configs[1]: every section owns [1, 3073) -- the whole-rows search
    (5120, 1023, 4),     # the last section owns [1, 2049)
    (8200, 1023, 3),
    (2300, 200, 4),      # sections of 3896 kept lags: row 3 partly owned
    (6200, 1200, 4),
    (10000, 300, 2),
]


def _close(a, b, det, same_carrier_stage=True):
    assert a["flags"].tolist() == b["flags"].tolist()
    for f in ("block_idx", "template_id", "carrier_bin", "corr_sample"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("carrier_offset", "carrier_energy", "carrier_noise"):     # the same kernels: identical
        if same_carrier_stage:
            assert np.array_equal(a[f], b[f], equal_nan=True), f
        else:
            np.testing.assert_allclose(a[f], b[f], rtol=2e-6, atol=2e-6, equal_nan=True)
    m = (a["flags"] & F.FLAG_CARRIER) != 0
    np.testing.assert_allclose(a["corr_energy"][m], b["corr_energy"][m], rtol=3e-6)
    np.testing.assert_allclose(a["corr_noise"][m], b["corr_noise"][m], rtol=3e-6)
    np.testing.assert_allclose(a["corr_offset"][det], b["corr_offset"][det], atol=5e-6)


@pytest.mark.parametrize("h,w,nsec", CASES)
def test_sectioned_records_equal_the_oracle_and_the_unsectioned_kernel(h, w, nsec):
    rng = np.random.default_rng(h * 3 + w)
    tpl = synth.gold_template(10, 2) if w == 1023 else np.sign(rng.normal(0, 1, w))
    lo, hi = onp.unique_window(N, h, w)
    secs = F.plan_sections(N, h, w)
    assert len(secs) == nsec
    edge = [lo, lo + 1, lo + 2, hi - 1, hi - 2, hi - 3]
    for s in secs[1:]:
        edge += [s["win_lo"] - 2, s["win_lo"] - 1, s["win_lo"], s["win_lo"] + 1]
    pos = np.array(edge * 2 + list(rng.integers(lo, hi, 48)))
    nb = len(pos)
    blocks, _ = synth.synth_blocks(rng, nb, N, tpl, (lo, hi), signal_frac=1.0, positions=pos,
                                   carrier_bins=(12.0, 100.0))
    # a fifth of the randomly placed blocks carry noise only (a carrier tone without the code reaches the correlate stage too)
    tone = np.exp(2j * np.pi * 40.3 * np.arange(N) / N) * 0.05
    for i in range(2 * len(edge), nb, 5):
        z = rng.normal(0, 0.02, N) + 1j * rng.normal(0, 0.02, N) + tone
        blocks[i] = synth.quantise_iq(z)
    eng = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=64)
    assert eng.sections() == (nsec, 4096)
    rec = eng.detect(blocks, np.arange(nb))[:, 0]
    uns = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=64, path="unsectioned")
    assert uns.sections() == (0, 0)
    ref = uns.detect(blocks, np.arange(nb))[:, 0]
    det = (ref["flags"] & F.FLAG_CORR) != 0
    _close(rec, ref, det)
    # the window test in every row of every section: the same arithmetic, equal byte for byte
    gen = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=64, path="generic_rows")
    assert gen.sections() == (nsec, 4096)
    assert gen.detect(blocks, np.arange(nb))[:, 0].tobytes() == rec.tobytes()
    # complex64 input (the carrier kernels transform the raw bytes of u8 input, passes_w8.hpp
    # fwd_pass1_pre: their float fields differ from the complex64 run's in the last bit), sectioned
    # against unsectioned and against the u8 run
    c64 = ((blocks[:32].astype(np.float32) - 127.4) / 128).view(np.complex64)
    rc = eng.detect(c64, np.arange(32))[:, 0]
    _close(rc, uns.detect(c64, np.arange(32))[:, 0], det[:32])
    _close(rc, rec[:32], det[:32], same_carrier_stage=False)
    rows = soak_util.run_oracle(blocks, N, h, tpl, THR, (7, 110), THR, procs=8, chunk=16)
    mism, worst, ties = soak_util.compare(rec, rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR)
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst, ties)
    assert not ties
    assert worst["offset"] <= 5e-6 and worst["energy"] <= 2e-5 and worst["noise"] <= 2e-5, worst
    found = set(rec["corr_sample"][(rec["flags"] & F.FLAG_CORR) != 0].tolist())
    assert {int(lo), int(hi - 1)} <= found                                   # peaks ON both window edges
    assert all(s["win_lo"] in found and s["win_lo"] - 1 in found for s in secs[1:])   # and on both sides of a seam
    for e in (eng, uns, gen):
        e.close()


def test_what_keeps_the_unsectioned_kernel():
    """...and that the handle SAYS so (thr_get_path_info -> Engine.path_info(), Detector.engine_path):
    a stddev term or an odd history costs a sixth of the throughput without any other sign."""
    tpl = synth.gold_template(10, 2)
    tpl4 = np.stack([synth.gold_template(10, g) for g in (2, 3, 4, 5)])
    for kw, want, why in [(dict(), (4, 4096), "sectioned"),
                          (dict(path="unsectioned"), (0, 0), "path"),
                          (dict(path="unsectioned_generic_rows"), (0, 0), "path"),
                          (dict(path="multipass"), (0, 0), "path")]:
        e = F.Engine(N, 4096, tpl, THR, (7, 110), THR, max_batch=8, **kw)
        assert e.sections() == want, kw
        info = e.path_info()
        assert info["why_unsectioned"] == why and (info["n_sections"], info["section_len"]) == want, info
        e.close()
    # several templates run sectioned too (ABI 9)
    e = F.Engine(N, 4096, tpl4, THR, (7, 110), THR, max_batch=8)
    info = e.path_info()
    assert e.sections() == (4, 4096) and info["n_templates"] == 4 and info["correlate_kernel"] == "k_correlate_4k"
    assert info["carrier_kernel"] == "k_carrier_pruned" and "4 sections of 4096" in info["text"]
    e.close()
    long_tpl = np.sign(np.random.default_rng(1).normal(0, 1, 4914))
    for args, why, rows in [((N, 4096, tpl, THR, (7, 110), (0, 15, 0.5)), "stddev", (-1, -1)),   # sums over every kept lag
                            ((N, 4920, long_tpl, THR, (7, 110), THR), "geometry", (0, 4)),        # a template longer than a section
                            ((N, 1100, tpl, THR, (7, 110), THR), "geometry", (0, 1)),             # a window of five sections
                            ((8192, 2048, tpl, THR, (7, 110), THR), "block_len", (-1, -1)),       # whole blocks sit in LDS
                            ((65536, 4096, long_tpl[:4094], THR, (7, 110), THR), "sectioned", (0, 3))]:
        e = F.Engine(*args, max_batch=8)
        info = e.path_info()
        assert info["why_unsectioned"] == why, (args[:2], info)
        assert info["rows"] == rows, (args[:2], info)
        if why != "sectioned":
            assert e.sections() == (0, 0) and "unsectioned: " in info["text"]
        else:
            assert e.sections() == (5, 16384) and info["correlate_kernel"] == "k_correlate_seg"
        e.close()
    e = F.Engine(N, 4096, tpl, THR, (7, 110), THR, max_batch=8, preshift_num=64)
    assert e.sections() == (0, 0) and e.path_info()["why_unsectioned"] == "variant"
    assert e.path_info()["correlate_kernel"] == "k_preshift"
    e.close()
    # the drop-in class carries it and logs it once
    import logging
    from thrifty_amd.detect import Detector, DetectorSettings
    records = []
    handler = logging.Handler()
    handler.emit = records.append
    log = logging.getLogger("thrifty_amd.detect")
    log.addHandler(handler)
    old = log.level
    log.setLevel(logging.INFO)
    try:
        det = Detector(DetectorSettings(N, 4096, 1023, THR, (7, 110), tpl, (0, 15, 0.5)), None)
    finally:
        log.removeHandler(handler)
        log.setLevel(old)
    assert det.engine_path["why_unsectioned"] == "stddev"
    assert len(records) == 1 and "stddev term" in records[0].getMessage()
    det.close()


@pytest.mark.parametrize("n_tpl", [2, 4])
def test_several_templates_sectioned_against_unsectioned_and_oracle(n_tpl):
    """BASELINE configs[4]: one forward transform per section, one product + inverse per template
    (k_correlate_4k<MULTI>).  Every template's burst in turn, peaks on the seams and the window edges."""
    rng = np.random.default_rng(60 + n_tpl)
    tpls = np.stack([synth.gold_template(10, 2 + i) for i in range(n_tpl)]).astype(np.float64)
    h = 4096
    lo, hi = onp.unique_window(N, h, 1023)
    plan = F.plan_sections(N, h, 1023)
    seams = [s["win_lo"] for s in plan[1:]]
    pos = [lo, hi - 1] + [p for s in seams for p in (s - 1, s)] + list(rng.integers(lo, hi, 40))
    nb = len(pos)
    blocks = np.empty((nb, 2 * N), dtype=np.uint8)
    for i, p in enumerate(pos):       # block i carries template i % n_tpl
        b, _ = synth.synth_blocks(rng, 1, N, tpls[i % n_tpl], (lo, hi), positions=np.array([p]), carrier_bins=(12.0, 100.0))
        blocks[i] = b[0]
    eng = F.Engine(N, h, tpls, THR, (7, 110), THR, max_batch=64)
    uns = F.Engine(N, h, tpls, THR, (7, 110), THR, max_batch=64, path="unsectioned")
    gen = F.Engine(N, h, tpls, THR, (7, 110), THR, max_batch=64, path="generic_rows")
    assert eng.sections() == (4, 4096) and uns.sections() == (0, 0)
    rec = eng.detect(blocks, np.arange(nb))
    ref = uns.detect(blocks, np.arange(nb))
    assert rec.shape == (nb, n_tpl)
    assert gen.detect(blocks, np.arange(nb)).tobytes() == rec.tobytes()
    for t in range(n_tpl):
        det = (ref[:, t]["flags"] & F.FLAG_CORR) != 0
        _close(rec[:, t], ref[:, t], det)
        assert np.array_equal(rec[:, t]["template_id"], np.full(nb, t))
        # the template a block carries is detected at the position it was put
        mine = np.arange(nb) % n_tpl == t
        assert np.all(det[mine]) and np.array_equal(rec[:, t]["corr_sample"][mine], np.array(pos)[mine])
    orc = onp.OracleDetector(N, h, tpls, THR, (7, 110), THR)
    for i in range(nb):
        want = orc.detect_u8(i, blocks[i])
        for t in range(n_tpl):
            r, w = rec[i, t], want[t]
            assert bool(r["flags"] & F.FLAG_CORR) == w.detected and r["carrier_bin"] == w.carrier.bin
            assert r["corr_sample"] == w.corr.sample
            assert abs(r["corr_energy"] - w.corr.energy) <= 2e-5 * w.corr.energy
            if w.detected:
                assert abs(r["corr_offset"] - w.corr.offset) <= 5e-6
    # complex64 input and a 1500-block batch (whole-block tickets) against the unsectioned kernel
    big = np.tile(blocks, (1500 // nb + 1, 1))[:1500]
    e2 = F.Engine(N, h, tpls, THR, (7, 110), THR, max_batch=1500)
    u2 = F.Engine(N, h, tpls, THR, (7, 110), THR, max_batch=1500, path="unsectioned")
    a, b = e2.detect(big, np.arange(1500)), u2.detect(big, np.arange(1500))
    for t in range(n_tpl):
        _close(a[:, t], b[:, t], (b[:, t]["flags"] & F.FLAG_CORR) != 0)
    c64 = ((blocks[:16].astype(np.float32) - 127.4) / 128).view(np.complex64)
    for t in range(n_tpl):
        rc, ru = eng.detect(c64, np.arange(16))[:, t], uns.detect(c64, np.arange(16))[:, t]
        _close(rc, ru, (ru["flags"] & F.FLAG_CORR) != 0)
    for e in (eng, uns, gen, e2, u2):
        e.close()


def test_stage_dumps_come_from_the_unsectioned_kernel_and_agree():
    """Detector.detect(yield_data=True) reads the shifted spectrum and the correlation of a block
    (detect.py:75-78): those launches keep k_correlate; their record must agree with the sectioned one."""
    rng = np.random.default_rng(9)
    tpl = synth.gold_template(10, 2)
    lo, hi = onp.unique_window(N, 4096, 1023)
    blocks, _ = synth.synth_blocks(rng, 8, N, tpl, (lo, hi))
    eng = F.Engine(N, 4096, tpl, THR, (7, 110), THR, max_batch=8)
    rec = eng.detect(blocks, np.arange(8))[:, 0]
    xhat, corr = eng.debug_stage(blocks, 0)
    mag = np.abs(corr[:, :N - 1023 + 1])
    assert np.array_equal(np.argmax(mag[:, lo:hi], axis=1) + lo, rec["corr_sample"])
    np.testing.assert_allclose(mag[np.arange(8), rec["corr_sample"]], rec["corr_energy"], rtol=3e-6)
    eng.close()


def test_raw_stream_framing_and_large_batches():
    """Overlapping blocks of a raw stream (block stride 2 (N - H) bytes, block_data.py:70-98) and a
    batch larger than one launch's resident workgroups: sectioned == unsectioned."""
    rng = np.random.default_rng(21)
    tpl = synth.gold_template(10, 2)
    h = 4096
    lo, hi = onp.unique_window(N, h, 1023)
    nb = 1500
    blocks, _ = synth.synth_blocks(rng, 64, N, tpl, (lo, hi), signal_frac=0.8)
    big = np.tile(blocks, (nb // 64 + 1, 1))[:nb]
    a = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=2048)
    b = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=2048, path="unsectioned")
    ra, rb = a.detect(big, np.arange(nb))[:, 0], b.detect(big, np.arange(nb))[:, 0]
    _close(ra, rb, (rb["flags"] & F.FLAG_CORR) != 0)
    assert ((ra["flags"] & F.FLAG_CORR) != 0).sum() > nb // 2
    # a raw stream: 40 overlapping blocks with a burst every 9000 samples
    ns = (N - h) * 40 + h
    z = rng.normal(0, 0.02, ns) + 1j * rng.normal(0, 0.02, ns)
    ook = 0.3 * (tpl + 1) / 2
    for p in range(3000, ns - 2000, 9000):
        z[p:p + 1023] += ook * np.exp(2j * np.pi * 33.4 * np.arange(p, p + 1023) / N)
    stream = synth.quantise_iq(z)
    sa, sb = a.detect_stream(stream), b.detect_stream(stream)
    assert len(sa) == 40 and ((sb[:, 0]["flags"] & F.FLAG_CORR) != 0).sum() >= 30
    _close(sa[:, 0], sb[:, 0], (sb[:, 0]["flags"] & F.FLAG_CORR) != 0)
    a.close()
    b.close()
    # a history that is not a multiple of 8: the blocks of the stream start on 4-byte boundaries (raw-stream
    # framing wants an even block_len - history_len), the sections' 16-byte sample fetches are unaligned
    h = 4098
    a = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=64)
    b = F.Engine(N, h, tpl, THR, (7, 110), THR, max_batch=64, path="unsectioned")
    assert a.sections() == (4, 4096)
    n2 = (N - h) * 40 + h
    sa, sb = a.detect_stream(stream[:2 * n2]), b.detect_stream(stream[:2 * n2])
    assert len(sa) == 40 and ((sb[:, 0]["flags"] & F.FLAG_CORR) != 0).sum() >= 30
    _close(sa[:, 0], sb[:, 0], (sb[:, 0]["flags"] & F.FLAG_CORR) != 0)
    a.close()
    b.close()


def test_random_sectioned_geometries_equal_the_unsectioned_kernel_and_the_oracle():
    """A seeded slice of tests/tools/fuzz_sections.py (400 configurations / 279 109 blocks on the round-5
    build: profiles/r05_soak.log): random history / template length with one to four sections, template
    kind, carrier window, thresholds, input format and batch sizes on both sides of the kernel's
    ticket switch -- exact fields equal the unsectioned kernel's and the oracle's."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_sections
    rng = np.random.default_rng(20260930)
    seen, templates = set(), set()
    for k in range(14):
        status, desc = fuzz_sections.one(rng, k, F, onp, synth, soak_util, 16)
        assert status == "ok", (desc, status)
        seen.add(desc.rsplit("sections=", 1)[1])
        templates.add(desc.split(" T=")[1].split()[0])
    assert len(seen) >= 3, seen
    assert templates & {"2", "4"} and "1" in templates, templates      # one template and several
