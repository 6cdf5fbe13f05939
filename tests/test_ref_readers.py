"""The host framing of `.card` text and raw streams (thrifty_amd.block_data, thr_frame_card)
against the REFERENCE's own native readers -- fastcard/card_reader.c + lib/base64.c and
fastcard/raw_reader.c, compiled from /root/reference by oracle/Makefile into oracle/_ref/ (the
`oracle/_ref` of the build contract; test infrastructure).  Skipped where that library has not
been built AND cannot be (no reference checkout); where the checkout exists -- or THRIFTY_REQUIRE_REF=1
says the library must have travelled with the tree -- its absence fails the suite."""
import base64
import io
import os

import numpy as np
import pytest

from oracle import ref_readers
from thrifty_amd import block_data

def ref_required():
    """The library may only be ABSENT where it cannot be built: with the reference checkout present
    (the build container: __graft_entry__.build() makes it) -- or wherever the caller says it must
    travel with the tree (THRIFTY_REQUIRE_REF=1) -- a missing oracle/_ref is a failure, not a skip."""
    stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", ".ref_expected")
    return (os.path.isdir("/root/reference/fastcard") or os.environ.get("THRIFTY_REQUIRE_REF") == "1"
            or os.path.exists(stamp))     # (the stamp __graft_entry__.build() leaves: it travels to the GPU box)


def test_the_reference_readers_are_built_wherever_they_can_be():
    if not ref_readers.available():
        assert not ref_required(), ("oracle/_ref/libfastcard_readers.so is missing although the reference "
                                    "checkout is here (or THRIFTY_REQUIRE_REF=1): run `make -C oracle` / "
                                    "__graft_entry__.build()")
        pytest.skip("oracle/_ref/libfastcard_readers.so not built and no reference checkout to build it from")


needs_ref = pytest.mark.skipif(not ref_readers.available() and not ref_required(),
                               reason="oracle/_ref/libfastcard_readers.so not built (make -C oracle)")


def _card_file(tmp_path, n, nblk, rng, with_comments=True):
    raws = rng.integers(0, 256, (nblk, 2 * n), dtype=np.uint8)
    lines = []
    if with_comments:
        lines.append("# fastcard capture, block size %d\n" % n)
    for i in range(nblk):
        lines.append(block_data.card_line(1500000000.25 + 0.125 * i, 7 + 3 * i, raws[i]))
        if with_comments and i == 2:
            lines.append("# a comment between records\n")
    path = tmp_path / "rx.card"
    path.write_text("".join(lines))
    return path, raws


@needs_ref
@pytest.mark.parametrize("n", [64, 4096, 16384])
def test_card_framing_and_decode_equal_the_native_reader(tmp_path, n):
    rng = np.random.default_rng(n)
    path, raws = _card_file(tmp_path, n, 9, rng)
    ref, rc = ref_readers.read_blocks(str(path), n, 0, card=True)
    assert rc == 1 and len(ref) == 9                   # clean end of file (card_reader.c:46-75)
    for i, (sec, usec, idx, data) in enumerate(ref):
        assert np.array_equal(data, raws[i])           # the reference's own base64 decode
    # classic reader (block_data.card_reader, host decode)
    with open(path, "r") as f:
        ours = list(block_data.card_reader(f))
    assert len(ours) == len(ref)
    for (ts, idx, blk), (sec, usec, ridx, data) in zip(ours, ref):
        assert idx == ridx and abs(ts - (sec + usec * 1e-6)) < 1e-9
        assert np.array_equal(np.asarray(blk.raw), data)
    # batch reader over the mapped file: framing by the engine library's thr_frame_card, payloads
    # decoded here with Python's base64 (the device decode is pinned to this in tests/test_gpu_card_ingest.py)
    with open(path, "rb") as f:
        cs = block_data.CardStream(f, n)
        stamps, idxs, text, offs = cs.next_batch(100)
        chars = cs.payload_chars
        got = [np.frombuffer(base64.b64decode(bytes(text[o:o + chars])), dtype=np.uint8) for o in offs]
    assert [int(v) for v in idxs] == [r[2] for r in ref]
    for ts, g, (sec, usec, _, data) in zip(stamps, got, ref):
        assert abs(ts - (sec + usec * 1e-6)) < 1e-9 and np.array_equal(g, data)
    assert cs.next_batch(100) is None


@needs_ref
def test_card_history_copy_of_the_native_reader(tmp_path):
    """card_reader.c copies the previous block's tail in front of every decode (card_reader.c:27-33)
    and then overwrites the whole block: a .card block is self-contained, which is why blocks shard
    over GPUs without halo exchange."""
    n = 256
    path, raws = _card_file(tmp_path, n, 4, np.random.default_rng(3), with_comments=False)
    with_hist, _ = ref_readers.read_blocks(str(path), n, 64, card=True)
    without, _ = ref_readers.read_blocks(str(path), n, 0, card=True)
    for a, b, raw in zip(with_hist, without, raws):
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[3], raw)


@needs_ref
@pytest.mark.parametrize("bad,code", [("short", -4), ("long", -5), ("meta", -2)])
def test_malformed_lines_are_refused_by_both(tmp_path, bad, code):
    n = 64
    raw = np.arange(2 * n, dtype=np.uint8)
    good = block_data.card_line(1.5, 1, raw)
    if bad == "short":
        line = good[:-9] + "\n"
    elif bad == "long":
        line = good[:-1] + "AAAA\n"
    else:
        line = "not-a-timestamp 3 " + good.split(" ", 2)[2]
    path = tmp_path / "bad.card"
    path.write_text(good + line)
    ref, rc = ref_readers.read_blocks(str(path), n, 0, card=True)
    assert len(ref) == 1 and rc == code                 # card_reader.c:55-72
    # like the native reader: the line before the bad one is delivered, then the error
    for force_py in (False, True):
        with open(path, "rb") as f:
            cs = block_data.CardStream(f, n)
            nb = cs._next_batch_py if force_py else cs.next_batch
            stamps, idxs, _, offs = nb(10)
            assert list(idxs) == [1] and stamps[0] == 1.5 and len(offs) == 1
            with pytest.raises(ValueError):
                nb(10)
    # (the classic card_reader follows the PYTHON reference, block_data.py:120-131, which decodes
    # whatever the line holds and leaves the length check to Detector.detect's assert)
    with open(path, "r") as f:
        rd = block_data.card_reader(f)
        assert next(rd)[1] == 1
        # (what it does with the bad line is the Python reference's business: b64decode may raise,
        # stop at the padding, or hand back a block of another length)


@needs_ref
@pytest.mark.parametrize("n,h", [(64, 16), (4096, 1024), (16384, 4920)])
def test_raw_stream_framing_equals_the_native_reader(tmp_path, n, h):
    """raw_reader.c:15-46: block i = the previous block's last `history` samples + n - history new
    ones.  Python's block_reader starts from an all-zero (0.0) history, fastcard from whatever the
    block was initialised with (RAWCONV_ZERO pairs): from the first block whose history is real
    samples on, the two -- and RawStream's framing, which the engine reads in place -- agree byte
    for byte."""
    rng = np.random.default_rng(n + h)
    new = n - h
    nblk = 7
    stream = rng.integers(0, 256, 2 * new * nblk + 10, dtype=np.uint8)     # (+ a ragged tail: dropped)
    path = tmp_path / "rx.bin"
    path.write_bytes(stream.tobytes())
    init = np.zeros(2 * n, dtype=np.uint8)
    ref, rc = ref_readers.read_blocks(str(path), n, h, card=False, initial=init)
    assert rc == 1 and len(ref) == nblk and [r[2] for r in ref] == list(range(nblk))
    lead = -(-h // new)                                   # blocks that still see initial history
    for i, (_, _, _, data) in enumerate(ref):
        lo = 2 * (new * (i + 1) - n)
        want = stream[max(lo, 0):2 * new * (i + 1)]
        assert np.array_equal(data[2 * n - len(want):], want)
        if i >= lead:
            assert len(want) == 2 * n
    # block_reader (host framing): blocks with their raw bytes attached once the history is real
    with open(path, "rb") as f:
        ours = list(block_data.block_reader(f, n, h))
    assert len(ours) == nblk
    for i, (_, idx, blk) in enumerate(ours):
        assert idx == i
        if i >= lead:
            assert np.array_equal(np.asarray(blk.raw), ref[i][3])
            assert np.array_equal(np.asarray(blk), block_data.raw_to_complex(ref[i][3]))
    # RawStream (what the engine frames on the device): u8 batches are windows of the byte stream
    with open(path, "rb") as f:
        rs = block_data.RawStream(f, n, h)
        seen = []
        while True:
            b = rs.next_batch(3)
            if b is None:
                break
            kind, stamps, idxs, data = b
            if kind == "u8":
                view = np.frombuffer(data, dtype=np.uint8)
                for j, idx in enumerate(idxs):
                    seen.append((int(idx), view[2 * new * j:2 * new * j + 2 * n].copy()))
            else:
                for j, idx in enumerate(idxs):
                    assert int(idx) < lead
    assert [i for i, _ in seen] == list(range(lead, nblk))
    for i, data in seen:
        assert np.array_equal(data, ref[i][3])
