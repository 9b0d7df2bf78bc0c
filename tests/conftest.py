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


def pytest_sessionfinish(session, exitstatus):
    """TTC_GUARD=<KiB> runs (tools/run_guarded_gpu_tests.sh): every libttc context scans the guard zones around its device buffers when it is
    closed (sentinel-tree-cover_amd/_lib.py, csrc/ttc_internal.h); the tally goes to gpurun_out/device_guard_report.txt and a violation fails the run"""
    if os.environ.get("TTC_GUARD", "0") in ("", "0"):
        return
    import gc
    gc.collect()
    try:
        from ttc import _lib
    except Exception:
        return
    st = _lib.GUARD_STATS
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "device_guard_report.txt"), "a") as f:
        f.write("TTC_GUARD=%s KiB before and after every device buffer the library owns; pytest exit status %s; contexts checked at close: %d; "
                "violations: %d\n" % (os.environ["TTC_GUARD"], exitstatus, st["contexts_checked"], len(st["violations"])))
        for v in st["violations"]:
            f.write("  VIOLATION " + v + "\n")
    if st["violations"]:
        session.exitstatus = 1
