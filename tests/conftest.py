import os
import sys

import pytest

# The oracle is OpenMP code; on a 256-thread host the fork/join cost of its small
# parallel regions dominates the small test problems.  Cap it for the test-suite
# (bench.py's cpu_baseline leg uses every core).
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
