import os
import sys

import numpy as np
import pytest
import torch

# the CPU oracle is slowest with one thread per logical CPU on big hosts (tools/cpu_threads_scan.py)
torch.set_num_threads(min(16, torch.get_num_threads()))

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def ops_kat():
    return load_golden("ops_kat.npz")


@pytest.fixture(scope="session")
def gen_tiny():
    return load_golden("gen_tiny.npz")
