import json
import os
import sys

import pytest

try:  # load torch's bundled ROCm runtime BEFORE libhypergen_hip.so pulls in /opt/rocm's: with the
    import torch  # noqa: F401  opposite order torch later reports "No HIP GPUs are available"
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
