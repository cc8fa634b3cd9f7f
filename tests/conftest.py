import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cores() -> int:
    """CPU cores this process may really use: min(affinity mask, cgroup cpu.max quota).  The GPU boxes expose 256
    logical CPUs and cap the container at 16 by cgroup; torch sizes its intra-op pool from the former, and the host
    oracle runs of the parity tests (128x256 fp32 + fp64, 100-200 s each) then crawl on 256 threads sharing 16 cores
    (round 6: 568 s for one test, 1,060 s for the suite against the driver's 1,200 s limit)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(usable_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
