import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import sgmcmc_oracle
    sgmcmc_oracle.load_c()
    return sgmcmc_oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no HIP device is visible")
    from pysgmcmc_amd import _lib
    _lib.lib()   # fail loudly if the extension is missing
    return torch.device("cuda:0")
