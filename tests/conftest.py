import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's ops are small: on the GPU box's 256 hardware threads (shared with other jobs) torch's default pool
    # makes them several times slower than 16 threads do (bench.py's cpu_baseline probes the same)
    try:
        import torch

        torch.set_num_threads(min(16, torch.get_num_threads()))
    except Exception:  # pragma: no cover
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a ROCm device: the `gpu`-marked tests are skipped (not failed), so a plain run shows
    only real CPU regressions.  With a device present nothing is skipped."""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (MI355X): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
