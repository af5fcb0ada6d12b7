import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle runs on libTorch's CPU kernels: keep its thread pool within the CPU share this process really has (a GPU box hands out
    # 16 cores of a much larger host; the default pool of one thread per host core spends its time in contention)
    import torch
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
