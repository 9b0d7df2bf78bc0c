import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


ORACLE_THREADS = 16      # torch CPU oracle: 16 threads are ~9x FASTER than the 128 torch picks on the 256-cpu GPU box
                         # (profiles/r03_cpu_threads.txt: 172^2 window 68 ms vs 598 ms)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver's GPU tier)")
    try:
        import torch
        torch.set_num_threads(min(ORACLE_THREADS, os.cpu_count() or 1))
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
