import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Every test gets a wall-clock bound (pytest-timeout, when installed): a rendezvous that never completes or a wedged subprocess fails
    ONE test after five minutes instead of hanging the suite (the 2-rank gloo tests and the full-size GPU tests finish in well under a minute)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    # torch's default intra-op thread count on the 256-logical-CPU GPU host (128) makes the CPU oracle several times SLOWER than
    # 16-32 threads do (tools/cpu_probe.py): cap it for every test process
    try:
        import torch
        torch.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:
        pass
