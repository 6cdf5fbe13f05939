"""Last in the GPU suite (the file name sorts behind every other test): the committed counter
figures must belong to the committed kernel sources."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_traffic_figures_belong_to_the_committed_kernel_sources():
    """`roofline.traffic` comes from profiles/hbm_traffic.json (PMC passes, scripts/profile_gpu.sh);
    the file names the hash of the kernel sources it was taken on.  A kernel edit without new
    counter passes must not reach the driver's line as `traffic_stale: true` unnoticed."""
    from thrifty_amd import build
    data = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    want = build.csrc_hash()
    for key in ("c2", "c2_t4", "c2_sparse", "c2_fullwin", "c3", "c3_t4", "c1", "c1_sparse"):
        assert key in data, "no counter passes for %s in profiles/hbm_traffic.json" % key
        assert data[key]["_source"]["csrc_sha16"] == want, (
            "%s: counters taken on sources %s, the tree is %s -- re-run scripts/profile_gpu.sh"
            % (key, data[key]["_source"]["csrc_sha16"], want))
