"""pytest configuration: the `gpu` marker and shared paths/fixtures."""
import json
import os
import sys

# The GPU boxes show far more cores than their CPU quota; OpenMP's default (one thread per visible core) then
# oversubscribes the oracle and UpdateWorld_CPU by 10x.  Results do not depend on the thread count.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Wall-clock assertions run after every correctness row, whatever the files are called: the driver runs the GPU
    suite with -x, and a slow box must not leave oracle / fixture / transport tests untested."""
    perf = [it for it in items if it.fspath.basename == "test_gpu_zz_perf.py"]
    if perf:
        items[:] = [it for it in items if it.fspath.basename != "test_gpu_zz_perf.py"] + perf


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    """golden(name) -> float32 array (n, 8) of a committed fixture file."""

    def load(name):
        return np.fromfile(os.path.join(GOLDEN, name), dtype=np.float32).reshape(-1, 8)

    return load
