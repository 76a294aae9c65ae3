import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_threads():
    """The GPU box gives one GPU's share of the host (16 cores) while PyTorch defaults to one thread per VISIBLE core (128 +): the CPU
    references (fp64 autograd of the C5 network, the PyTorch-fed MCTS) then fight over the share — the same test took 2 s on one box and
    30 s on another.  Cap the pool at the share."""
    try:
        import torch

        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except ImportError:
        pass
    yield
