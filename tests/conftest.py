import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    from oracle import host_cores
    torch.set_num_threads(min(host_cores(), 16))     # the CPU oracle must not oversubscribe the box's CPU share


@pytest.fixture(scope="session")
def edrl():
    import edrl_amd
    return edrl_amd


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def switches():
    """Set EDRL_* library switches for one test (environment + edrl_config_reload(): the library reads its environment once);
    the previous values are restored and re-read afterwards.  Usage: switches(EDRL_BF16_V3="2")."""
    import edrl_amd
    saved = {}

    def set_(**kw):
        for k in kw:
            saved.setdefault(k, os.environ.get(k))
        edrl_amd._lib.set_switches(**kw)

    yield set_
    if saved:
        edrl_amd._lib.set_switches(**saved)
