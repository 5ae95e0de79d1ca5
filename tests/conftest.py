import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def sub_state(npz, prefix):
    """state-dict slice `prefix.*` of a fixture, with the prefix stripped."""
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def tiny_hps():
    import json
    with open(os.path.join(GOLDEN, "tiny_hps.json")) as f:
        return json.load(f)
