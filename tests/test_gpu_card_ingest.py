"""GPU tests of the on-device .card ingest: base64 decode kernel +
CardStream batch reader, against the host decode path and the reference goldens."""
import io

import numpy as np
import pytest

from thrifty_amd import _native as F
from thrifty_amd import block_data
from thrifty_amd.block_data import CardStream
from thrifty_amd.detect import Detector

from test_gpu_parity import check_against_golden, engine_for
from test_gpu_detector_api import assert_toad_close, card_text, settings_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c2", "small"])
def test_device_decode_equals_host_decode(golden, name):
    g = golden(name)
    text = card_text(g).encode()
    cs = CardStream(io.BytesIO(text), int(g["block_len"]))
    stamps, idxs, buf, offs = cs.next_batch(1000)
    assert np.array_equal(idxs, g["block_idx"]) and stamps[0] == 1000.0
    eng = engine_for(g, max_batch=5)            # forces several chunks per call
    rec_card = eng.detect_card(buf, offs, idxs)[:, 0]
    rec_host = eng.detect(g["blocks"], g["block_idx"])[:, 0]
    for f in rec_host.dtype.names:
        assert np.array_equal(rec_card[f], rec_host[f]), f      # bit-identical
    check_against_golden(rec_card, g)


def test_detector_over_card_stream_matches_reference_toad(golden):
    g = golden("small")
    det = Detector(settings_of(g), CardStream(io.BytesIO(str(g["card_text"]).encode()), 4096,
                                               chunk_bytes=50000), rxid=3, batch_size=4)
    lines = [res.serialize() for detected, res in det if detected]
    assert_toad_close(lines, g["card_toad"])


def test_bad_payloads_are_rejected(golden):
    g = golden("c2")
    eng = engine_for(g)
    line = block_data.card_line(1.0, 7, g["blocks"][0])
    bad = line.replace("A", "!", 1) if "A" in line.split(" ")[2] else line[:40] + "!" + line[41:]
    cs = CardStream(io.BytesIO(bad.encode()), 16384)
    _, idxs, buf, offs = cs.next_batch(4)
    with pytest.raises(F.NativeError, match="base64"):
        eng.detect_card(buf, offs, idxs)
    # padding in the middle of a payload is invalid too
    pay = line.split(" ")[2]
    mid = " ".join(line.split(" ")[:2]) + " " + pay[:100] + "=" + pay[101:]
    _, idxs, buf, offs = CardStream(io.BytesIO(mid.encode()), 16384).next_batch(4)
    with pytest.raises(F.NativeError):
        eng.detect_card(buf, offs, idxs)
    # wrong payload length is caught by the host-side framing
    with pytest.raises(ValueError, match="payload"):
        CardStream(io.BytesIO((line[:-9] + "\n").encode()), 16384).next_batch(4)
    # and the engine still works afterwards
    _, idxs, buf, offs = CardStream(io.BytesIO(line.encode()), 16384).next_batch(4)
    rec = eng.detect_card(buf, offs, idxs)[:, 0]
    assert rec[0]["block_idx"] == 7 and rec[0]["corr_sample"] == g["sample"][0]
