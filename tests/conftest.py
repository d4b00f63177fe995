import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_test_tools():
    """Builds the oracle and the CPU model (host compilers only; no GPU involved)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "model")])


def has_gpu():
    """Asked of the product library's own HIP runtime (torch bundles a different
    libamdhip64; the GPU processes of this repo do not mix the two)."""
    try:
        from yadcc_amd import binding
        return binding.device_count() > 0
    except Exception:
        return False
