import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_blocks(g):
    """Input blocks of a fixture: its own, or those of the fixture it names as `src` (the
    interpolator variants of the preshift fixtures share the base fixture's inputs)."""
    return g["blocks"] if "blocks" in g.files else load_golden(str(g["src"]))["blocks"]


@pytest.fixture(scope="session")
def golden():
    return load_golden
