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


@pytest.fixture(scope="session", autouse=True)
def _gpu_memory_share():
    """JB_TEST_GPU_MEM_FRACTION=f: cap this process's share of the GPU's memory (torch's allocator raises instead of
    exhausting the device) -- for running the GPU suite as several pytest-xdist workers on one GPU (`-n 2`, f = 0.45)."""
    f = os.environ.get("JB_TEST_GPU_MEM_FRACTION")
    if f:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_per_process_memory_fraction(float(f), 0)
    yield
