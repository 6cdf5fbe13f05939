"""GPU parity: the HIP engine (through the C ABI) against the reference-generated
golden fixtures and against the CPU oracle on fresh seeded inputs.

Tolerances (BASELINE.json north_star): SoA sample index and carrier bin
bit-exact; sub-sample offset and correlation energy within 1e-4; the rest as
SURVEY.md section 8(c) proposes.
"""
import numpy as np
import pytest

from oracle import thrifty_np as onp
from thrifty_amd import _native, synth

pytestmark = pytest.mark.gpu

F = _native

TOL_REL = 2e-5      # energies / noises (BASELINE asks for 1e-4; measured <= 2e-6)
TOL_OFF = 5e-6      # absolute, sub-sample offset (|offset| <= 0.6; BASELINE 1e-4, measured <= 1e-6)
TOL_COFF = 2e-4     # absolute, carrier sub-bin offset (same solver as the reference -- MINPACK
                    # lmdif -- fed float32 magnitudes that differ in the last digit)


def engine_for(g, templates=None, max_batch=64, path="auto"):
    tpl = g["template"] if templates is None else templates
    return F.Engine(int(g["block_len"]), int(g["history_len"]), tpl,
                    tuple(g["carrier_thresh"]), tuple(int(v) for v in g["carrier_window"]),
                    tuple(g["corr_thresh"]), max_batch=max_batch, path=path)


def check_against_golden(rec, g):
    nb = len(g["blocks"])
    assert rec.shape == (nb,)
    for i in range(nb):
        r = rec[i]
        assert r["block_idx"] == g["block_idx"][i]
        assert r["carrier_bin"] == g["cbin"][i], i
        if g["index_error"][i]:
            assert r["flags"] & F.FLAG_INDEX_ERROR
            continue
        assert bool(r["flags"] & F.FLAG_CARRIER) == bool(g["carrier_det"][i]), i
        np.testing.assert_allclose(r["carrier_energy"], g["cenergy"][i], rtol=TOL_REL)
        np.testing.assert_allclose(r["carrier_noise"], g["cnoise"][i], rtol=TOL_REL)
        if not g["carrier_det"][i]:
            assert r["corr_sample"] == -1
            continue
        np.testing.assert_allclose(r["carrier_offset"], g["coff"][i], atol=TOL_COFF, rtol=0)
        assert r["corr_sample"] == g["sample"][i], i
        assert bool(r["flags"] & F.FLAG_CORR) == bool(g["det"][i]), i
        np.testing.assert_allclose(r["corr_energy"], g["energy"][i], rtol=TOL_REL)
        np.testing.assert_allclose(r["corr_noise"], g["noise"][i], rtol=TOL_REL)
        np.testing.assert_allclose(r["corr_offset"], g["soff"][i], atol=TOL_OFF, rtol=TOL_REL)


@pytest.mark.parametrize("name", ["c2", "c2_negwin", "c2_straddle", "c2_stddev", "c2_fullwin", "c1"])
def test_records_match_reference_golden(golden, name):
    g = golden(name)
    eng = engine_for(g)
    rec = eng.detect(g["blocks"], g["block_idx"])
    check_against_golden(rec[:, 0], g)
    # complex64 input path (a `Signal` that was already converted) gives the same records
    rec2 = eng.detect(np.stack([onp.iq_u8_to_c64(b) for b in g["blocks"]]), g["block_idx"])
    check_against_golden(rec2[:, 0], g)


def test_multi_template_matches_reference_golden(golden):
    gs = [golden("c5_tx%d" % i) for i in range(4)]
    tpls = np.stack([g["template"] for g in gs]).astype(np.float64)
    eng = engine_for(gs[0], templates=tpls)
    rec = eng.detect(gs[0]["blocks"], gs[0]["block_idx"])
    assert rec.shape == (12, 4)
    for t, g in enumerate(gs):
        assert np.all(rec[:, t]["template_id"] == t)
        check_against_golden(rec[:, t], g)


def test_forward_fft_matches_numpy(golden):
    g = golden("c2")
    eng = engine_for(g)
    spec = eng.debug_fft(g["blocks"][:8])
    for i in range(8):
        ref = np.fft.fft(onp.iq_u8_to_c64(g["blocks"][i]).astype(np.complex128))
        err = np.linalg.norm(spec[i] - ref) / np.linalg.norm(ref)
        assert err < 1e-6, err


def test_stage_dumps_match_oracle(golden):
    """yield_data intermediates (detect.py:75-78): shifted spectrum and correlation."""
    g = golden("c2")
    eng = engine_for(g)
    orc = onp.OracleDetector(16384, 4096, g["template"], (0, 15, 0), (7, 110), (0, 15, 0))
    xhat, corr = eng.debug_stage(g["blocks"][:6])
    for i in range(6):
        (res,), ((xh, co),) = orc.detect_u8(0, g["blocks"][i], want_data=True)
        assert xh is not None
        e1 = np.linalg.norm(xhat[i] - xh) / np.linalg.norm(xh)
        e2 = np.linalg.norm(corr[i][:len(co)] - co) / np.linalg.norm(co)
        assert e1 < 5e-6 and e2 < 5e-6, (e1, e2)


def test_fresh_blocks_match_oracle():
    """Seeded blocks never seen by the fixtures: HIP vs oracle, incl. batching > grid."""
    rng = np.random.default_rng(777)
    n, h = 16384, 4096
    tpl = synth.gold_template(10, 7)
    win = onp.unique_window(n, h, len(tpl))
    blocks, _ = synth.synth_blocks(rng, 40, n, tpl, win, signal_frac=0.8)
    orc = onp.OracleDetector(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0))
    eng = F.Engine(n, h, tpl, (0, 15, 0), (7, 110), (0, 15, 0), max_batch=16)
    idx = np.arange(40) * 7 + 3
    rec = eng.detect(blocks, idx)[:, 0]
    for i in range(40):
        (res,) = orc.detect_u8(int(idx[i]), blocks[i])
        r = rec[i]
        assert r["carrier_bin"] == res.carrier.bin
        assert bool(r["flags"] & F.FLAG_CARRIER) == res.carrier.detected
        if res.carrier.detected:
            assert r["corr_sample"] == res.corr.sample
            assert bool(r["flags"] & F.FLAG_CORR) == res.corr.detected
            np.testing.assert_allclose(r["corr_energy"], res.corr.energy, rtol=TOL_REL)
            np.testing.assert_allclose(r["corr_offset"], res.corr.offset, atol=TOL_OFF, rtol=TOL_REL)
            np.testing.assert_allclose(r["carrier_offset"], res.carrier.offset, atol=TOL_COFF)


def test_device_resident_path_and_compaction(golden):
    torch = pytest.importorskip("torch")
    g = golden("c2")
    eng = engine_for(g)
    dev = torch.device("cuda:0")
    blocks = torch.from_numpy(g["blocks"]).to(dev)
    idx = torch.from_numpy(g["block_idx"]).to(dev)
    out = torch.zeros(len(g["blocks"]) * 64, dtype=torch.uint8, device=dev)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()     # (torch's default stream handle is 0 == "the engine's own stream")
    eng.profile_enable(True)
    eng.detect_device(blocks.data_ptr(), F.THR_IN_U8, len(g["blocks"]), out.data_ptr(), idx.data_ptr())
    torch.cuda.synchronize()
    rec = out.cpu().numpy().view(F.RECORD_DTYPE)
    check_against_golden(rec, g)
    prof = eng.profile_read()
    assert all(cnt == 1 and ms > 0 for k, (ms, cnt) in prof.items() if not k.endswith("(small batches)")), prof   # (slot 4: long blocks only)
    kept = torch.zeros_like(out)
    torch.cuda.synchronize()
    n_kept = eng.compact_device(out.data_ptr(), len(g["blocks"]), kept.data_ptr())
    krec = kept.cpu().numpy().view(F.RECORD_DTYPE)[:n_kept]
    want = rec[(rec["flags"] & F.FLAG_CORR) != 0]
    assert n_kept == len(want) == int(g["det"].sum())
    assert np.array_equal(krec["block_idx"], want["block_idx"])
    assert np.array_equal(krec["corr_sample"], want["corr_sample"])
