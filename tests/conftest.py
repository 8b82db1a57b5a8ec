import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def cpu_share():
    """CPUs this process may really use: the cgroup quota when there is one (the GPU box exposes 256
    logical CPUs but grants 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure).  A few threads only: the dense restatement opens one
    OpenMP region per group and is dominated by fork/join cost on many-core hosts."""
    from oracle import Oracle
    o = Oracle()
    o.set_num_threads(min(4, o.num_threads()))
    return o


@pytest.fixture(scope="session")
def gpu_core():
    """One Core on cuda:0.  Fails loudly (no skip, no fallback) when the extension or GPU is missing."""
    from msweep_amd.core import Core
    c = Core(0)
    yield c
    c.close()


def dense_from_csr(p, lut, zi=0.01):
    G = len(p["group_sizes"])
    rp = p["rowptr"].astype(np.int64)
    E = len(rp) - 1
    L = np.full((G, E), np.log(zi))
    rows = np.repeat(np.arange(E), np.diff(rp))
    L[p["grp"], rows] = lut[p["grp"], p["cnt"]]
    return L


def lutidx_of(p, lut):
    return (p["grp"].astype(np.uint32) * lut.shape[1] + p["cnt"]).astype(np.uint32)
