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


# tests of modules whose `ctx` fixture runs in both input forms ("ascii", "packed") but that never touch a sequence: the
# second form would repeat them unchanged
_NO_SEQUENCE = ("test_dist", "test_hamming", "test_binarize", "test_hv_encode", "test_wyrng", "test_ani_golden",
                "test_symmetric_large", "test_dev_api", "test_random_thresholded")


def pytest_collection_modifyitems(config, items):
    keep = []
    for it in items:
        cs = getattr(it, "callspec", None)
        if cs is not None and cs.params.get("ctx") == "packed" and getattr(it, "originalname", it.name).startswith(_NO_SEQUENCE):
            continue
        keep.append(it)
    items[:] = keep


# ANI comparisons with the CPU oracle (and between kernel variants): EQUAL since the device evaluates glibc's logf algorithm
# (hyper-gen_amd/csrc/hg_logf.h; tests/test_gpu_ani_exact.py sweeps it against the host's logf on every float in (0, 1]).
# BASELINE.json's north_star allows 1e-4; the tests do not use the allowance.  (Comparisons with the float64 / torch.log
# models of tests/test_gpu_fullsize.py keep their own 1e-4: those models are not the reference's arithmetic.)
ANI_TOL = 0.0


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
