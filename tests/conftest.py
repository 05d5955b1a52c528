import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cpus():
    """Scheduler affinity cut down by the cgroup CPU quota (the GPU boxes give a 16-cpu share of a 256-thread host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    # The CPU oracle (fp32 torch) runs inside many GPU tests.  torch sizes its thread pool by the HOST's logical cpus: 128 threads on a
    # 16-cpu share ran the oracle ~10x slower than 16 threads do (bench.py's cpu_baseline met the same thing in round 2).
    import torch
    torch.set_num_threads(max(1, min(_usable_cpus(), 32)))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "real_curve: the 30-step loss curve at the real DeiT widths (deselect with -m 'gpu and not real_curve')")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
