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
    """The C oracle (test infrastructure; never imported by the product)."""
    from oracle import oracle_c

    oracle_c.load()
    return oracle_c


@pytest.fixture(scope="session")
def ctx():
    """A GPU context; fails loudly if the HIP library or the device is missing."""
    import dashing_amd

    try:  # bring torch's HIP runtime up first: some tests hand torch device buffers to the library
        import torch

        torch.cuda.init()
    except Exception:
        pass
    c = dashing_amd.Context(0)
    yield c
    c.close()
