"""GPU parity of the long-block paths: block_len = 4 x 16384 (BASELINE config C3) and 2 x 16384,
against the reference goldens, the oracle, and each other -- "auto" = carrier stage of
csrc/detect_long.hip + the correlate stage as overlap-save sections of the 16384-point kernel
(csrc/detect_seg.hip), "unsectioned" = the decimated long transform pair of detect_long.hip (what
"auto" itself uses for stage dumps and for templates too long to section), "multipass" = the
generic pipeline."""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native as F
from thrifty_amd import block_data, synth

from test_gpu_parity import check_against_golden, engine_for

pytestmark = pytest.mark.gpu


PATHS = ["auto", "unsectioned"]


def test_c3_every_path_matches_the_golden(golden):
    g = golden("c3")
    recs = {}
    for path in ("auto", "unsectioned", "multipass"):
        recs[path] = engine_for(g, max_batch=8, path=path).detect(g["blocks"], g["block_idx"])[:, 0]
        check_against_golden(recs[path], g)
    for path in ("unsectioned", "multipass"):
        assert np.array_equal(recs["auto"]["corr_sample"], recs[path]["corr_sample"])
        assert np.array_equal(recs["auto"]["flags"], recs[path]["flags"])
        np.testing.assert_allclose(recs["auto"]["corr_energy"], recs[path]["corr_energy"], rtol=2e-5)
        np.testing.assert_allclose(recs["auto"]["corr_offset"], recs[path]["corr_offset"], atol=5e-6)
    # (the carrier stage is shared by "auto" and "unsectioned": identical carrier halves)
    for c in ("carrier_bin", "carrier_offset", "carrier_energy", "carrier_noise"):
        assert np.array_equal(recs["auto"][c], recs["unsectioned"][c])


def test_c3_stage_dumps(golden):
    g = golden("c3")
    eng = engine_for(g, max_batch=4)
    spec = eng.debug_fft(g["blocks"][:2])
    for i in range(2):
        ref = np.fft.fft(block_data.raw_to_complex(g["blocks"][i]).astype(np.complex128))
        assert np.linalg.norm(spec[i] - ref) / np.linalg.norm(ref) < 2e-6
    orc = onp.OracleDetector(65536, 4096, g["template"], (0, 15, 0), (7, 110), (0, 15, 0))
    xhat, corr = eng.debug_stage(g["blocks"][:2])
    for i in range(2):
        (res,), ((xh, co),) = orc.detect_u8(0, g["blocks"][i], want_data=True)
        assert np.linalg.norm(xhat[i] - xh) / np.linalg.norm(xh) < 5e-6
        assert np.linalg.norm(corr[i][:len(co)] - co) / np.linalg.norm(co) < 5e-6


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("n,bits,sps,cwin,cthr,h", [
    (32768, 11, 1.0, (7, 110), (0, 15, 0), 4096),          # R0 = 2; three sections
    (32768, 11, 2.0, (7, 110), (0, 15, 0), 4200),          # ... unequal section windows
    (65536, 11, 2.0, (-300, -20), (0, 15, 0), 4096),       # negative-bin window
    (65536, 11, 2.0, (0, -1), (100.0, 5.0, 2.0), 4096),    # full window + stddev terms
    (65536, 11, 2.0, (0, -1), (100.0, 5.0, 2.0), 6001),    # ... sections with ragged windows
    (65536, 10, 1.0, (7, 110), (0, 15, 0), 1500),          # short template: sections nearly disjoint
    (65536, 11, 4.5, (7, 110), (0, 15, 0), 9300),          # 9211-sample template: eight sections
    (65536, 11, 4.6, (7, 110), (0, 15, 0), 9500),          # 9416 samples: too long to section ("auto" falls back)
])
def test_long_blocks_match_oracle(n, bits, sps, cwin, cthr, h, path):
    tpl = synth.gold_template(bits, 3, sps)
    if path == "auto":
        assert bool(F.plan_sections(n, h, len(tpl))) == (len(tpl) <= 9361)
    win = onp.unique_window(n, h, len(tpl))
    rng = np.random.default_rng(n + len(tpl))
    lo, hi = (-250.0, -30.0) if cwin[0] < 0 and cwin[1] < 0 else (10.0, 100.0)
    blocks, _ = synth.synth_blocks(rng, 5, n, tpl, win, signal_frac=0.8, carrier_bins=(lo, hi))
    eng = F.Engine(n, h, tpl, cthr, cwin, (50.0, 8.0, 3.0) if cthr[2] else (0, 15, 0), max_batch=3, path=path)
    orc = onp.OracleDetector(n, h, tpl, cthr, cwin, (50.0, 8.0, 3.0) if cthr[2] else (0, 15, 0))
    idx = np.arange(5) * 3 + 1
    rec = eng.detect(blocks, idx)[:, 0]
    for i in range(5):
        (res,) = orc.detect_u8(int(idx[i]), blocks[i])
        r = rec[i]
        assert r["carrier_bin"] == res.carrier.bin
        assert bool(r["flags"] & F.FLAG_CARRIER) == res.carrier.detected
        np.testing.assert_allclose(r["carrier_energy"], res.carrier.energy, rtol=1e-4)
        np.testing.assert_allclose(r["carrier_noise"], res.carrier.noise, rtol=1e-4)
        if res.carrier.detected:
            assert r["corr_sample"] == res.corr.sample
            assert bool(r["flags"] & F.FLAG_CORR) == res.corr.detected
            np.testing.assert_allclose(r["corr_energy"], res.corr.energy, rtol=1e-4)
            np.testing.assert_allclose(r["corr_noise"], res.corr.noise, rtol=1e-4)
            np.testing.assert_allclose(r["corr_offset"], res.corr.offset, atol=1e-4)
            np.testing.assert_allclose(r["carrier_offset"], res.carrier.offset, atol=1e-3)


@pytest.mark.parametrize("path", PATHS)
def test_long_multi_template_and_c64_input(path):
    n, h = 65536, 4096
    tpls = np.stack([synth.gold_template(11, i, 2.0) for i in (2, 3, 4)]).astype(np.float64)
    win = onp.unique_window(n, h, tpls.shape[1])
    rng = np.random.default_rng(77)
    parts = [synth.synth_blocks(rng, 2, n, t, win)[0] for t in tpls]
    blocks = np.concatenate(parts)
    eng = F.Engine(n, h, tpls, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=4, path=path)
    rec = eng.detect(blocks)
    rec_c64 = eng.detect(np.stack([block_data.raw_to_complex(b) for b in blocks]))
    for t in range(3):
        orc = onp.OracleDetector(n, h, tpls[t], (0, 15, 0), (7, 110), (0, 15, 0))
        for i in range(6):
            (res,) = orc.detect_u8(i, blocks[i])
            for r in (rec[i, t], rec_c64[i, t]):
                assert r["corr_sample"] == res.corr.sample
                assert bool(r["flags"] & F.FLAG_CORR) == res.corr.detected
                np.testing.assert_allclose(r["corr_energy"], res.corr.energy, rtol=1e-4)
                np.testing.assert_allclose(r["corr_offset"], res.corr.offset, atol=1e-4)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("n,n_tpl,fmt_c64,total", [
    (65536, 1, False, 900),     # fused kernel, several blocks per workgroup (combine_own, PeakTail carry-over)
    (65536, 1, True, 600),      # ... complex64 input
    (32768, 1, False, 1200),    # R0 = 2
    (65536, 3, False, 600),     # several templates: rows combined from memory after the template loop
])
def test_large_batches_equal_small_batches(n, n_tpl, fmt_c64, total, path):
    """("auto": thousands of (block, section) items walked by 256 workgroups through the dynamic
    cursor against batches of three blocks -- one item per workgroup; "unsectioned":)  One sub-batch with more carrier-positive blocks than workgroups takes the fused kernel with
    several blocks per workgroup (next-block prefetch, the peak written one barrier into the next
    block, the workgroup's last block after the loop); batches of <= 3 blocks take one block per
    workgroup or the two-kernel form.  Same arithmetic, so the records must be byte-identical --
    and the small-batch form is the one the oracle tests above pin."""
    h = 4096
    tpls = np.stack([synth.gold_template(11, 2 + i, 2.0 if n == 65536 else 1.0)
                     for i in range(n_tpl)]).astype(np.float64)
    win = onp.unique_window(n, h, tpls.shape[1])
    rng = np.random.default_rng(n + n_tpl + total)
    distinct = 48
    base, _ = synth.synth_blocks(rng, distinct, n, tpls[0], win, signal_frac=0.75)
    order = rng.integers(0, distinct, total)          # carrier-less blocks scattered through the batch
    blocks = base[order]
    inp = np.stack([block_data.raw_to_complex(b) for b in base])[order] if fmt_c64 else blocks
    idx = np.arange(total) * 2 + 5
    tp = tpls if n_tpl > 1 else tpls[0]
    big = F.Engine(n, h, tp, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=total, path=path).detect(inp, idx)
    small = F.Engine(n, h, tp, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=3, path=path).detect(inp, idx)
    assert int(((big["flags"][:, 0] & F.FLAG_CARRIER) != 0).sum()) > 300    # (more than 256 workgroups' worth)
    assert big.tobytes() == small.tobytes()
    # and a spot check against the oracle on the distinct blocks
    orc = onp.OracleDetector(n, h, tpls[0], (0, 15, 0), (7, 110), (0, 15, 0))
    for j in range(6):
        i = int(np.flatnonzero(order == j)[0]) if (order == j).any() else None
        if i is None:
            continue
        (res,) = orc.detect_u8(int(idx[i]), blocks[i])
        r = big[i, 0]
        assert bool(r["flags"] & F.FLAG_CARRIER) == res.carrier.detected
        if res.carrier.detected:
            assert r["corr_sample"] == res.corr.sample
            np.testing.assert_allclose(r["corr_energy"], res.corr.energy, rtol=1e-4)


def test_c3_benchmark_shape_one_full_sub_batch_against_the_oracle():
    """BASELINE configs[2] at the shape bench.py runs it: ONE launch batch of 16384 blocks (the
    long path's whole sub-batch: every workgroup of the fused kernel walks 64 blocks, the carrier
    stage its 16384 x 4 decimated sequences) holding 2048 FRESH blocks of bench.py's generator,
    each at eight scattered slots.  Every copy must give the same record (slot / workgroup
    independence), and the 2048 distinct blocks are checked against the CPU oracle: carrier bin,
    both verdicts and the SoA sample index exact, energies 2e-5, sub-sample offset 5e-6."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench
    import soak_util
    n, h, fresh, slots = 65536, 4096, 2048, 16384
    dev = torch.device("cuda", 0)
    tpl = synth.gold_template(11, 2, 2.0).astype(np.float64)
    win = onp.unique_window(n, h, len(tpl))
    gen = torch.Generator(device=dev)
    gen.manual_seed(bench.SEED + 33)
    base = bench.synth_on_device(torch, dev, gen, fresh, n, tpl, win, 0.9)
    order = torch.cat([torch.randperm(fresh, generator=gen, device=dev) for _ in range(slots // fresh)])
    data = base[order]                                    # 2 GiB: one full sub-batch
    idx = torch.arange(1000, 1000 + slots, dtype=torch.int64, device=dev)
    rec_d = torch.zeros((slots, 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=slots)
    eng.detect_device(data.data_ptr(), F.THR_IN_U8, slots, rec_d.data_ptr(), idx.data_ptr())
    eng.sync()
    rec = rec_d.cpu().numpy().view(F.RECORD_DTYPE).reshape(-1)
    assert np.array_equal(rec["block_idx"], np.arange(1000, 1000 + slots))
    order_h = order.cpu().numpy()
    firsts = np.full(fresh, -1)
    firsts[order_h[::-1]] = np.arange(slots)[::-1]        # first slot of every distinct block
    cols = [c for c in F.RECORD_DTYPE.names if c != "block_idx"]
    ref = rec[firsts][order_h]                            # the first copy's record, per slot
    for c in cols:
        assert np.array_equal(rec[c], ref[c]), c          # all eight copies identical, bit for bit
    blocks = base.cpu().numpy()
    rows = soak_util.run_oracle(blocks, n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), chunk=16)
    mism, worst, ties = soak_util.compare(rec[firsts], rows, blocks, F.FLAG_CARRIER, F.FLAG_CORR)
    assert sum(1 for r in rows if r is not None and r[5]) > 0.8 * fresh
    assert mism == dict(bin=0, carrier=0, sample=0, det=0, index_error=0), (mism, worst, ties)
    assert len(ties) <= 1, ties
    assert worst["offset"] <= 5e-6 and worst["energy"] <= 2e-5 and worst["noise"] <= 2e-5, worst
    assert worst["car_off"] <= 2e-4 and worst["car_energy"] <= 2e-5, worst
