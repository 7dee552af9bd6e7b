"""The C++ host shim (theia::BundleAdjuster on the C ABI): problem-semantics checks
on CPU, end-to-end BundleAdjustReconstruction / partial / view / track BA on GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

EXE = os.path.join(ROOT, "tests", "cpp", "test_host_shim")


def _run(mode):
    entry.build_engine()
    entry.build_host_shim()
    p = subprocess.run([EXE, mode], capture_output=True, text=True, timeout=600)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr


def test_host_shim_semantics_cpu():
    _run("cpu")


@pytest.mark.gpu
def test_host_shim_end_to_end_gpu():
    _run("gpu")
