import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def map_v1():
    d = np.load(os.path.join(GOLDEN, "map_v1.npz"))
    return d["mean"], d["cov"]


@pytest.fixture(scope="session")
def map_v2():
    d = np.load(os.path.join(GOLDEN, "map_v2.npz"))
    return d["mean"], d["cov"]


@pytest.fixture(scope="session")
def gt_sync():
    return np.load(os.path.join(GOLDEN, "gt_sync.npz"))


@pytest.fixture
def opt(gpu):
    """opt(name, value): set a context option (gl_ctx_set_option) for this test only."""
    ctx = gpu[1]
    saved = {}

    def set_(name, value):
        saved.setdefault(name, ctx.get_option(name))
        ctx.set_option(name, value)
    yield set_
    for k, v in saved.items():
        ctx.set_option(k, v)


@pytest.fixture(scope="session")
def gpu():
    """(torch, Context) on cuda:0 -- GPU tests fail loudly if the HIP library is absent."""
    import torch
    assert torch.cuda.is_available(), "GPU test collected without a GPU"
    import gmmloc_amd
    ctx = gmmloc_amd.Context(0)
    return torch, ctx
