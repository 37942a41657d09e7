import os
import sys

# The oracle's per-point float64 linear algebra (k <= 96) runs on the host: on a box that shows hundreds of cores to a
# process entitled to sixteen, every tiny eigh / matmul fanned out over all of them takes 100x longer than on one core.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(4)
    except Exception:
        pass
    # GPU sessions: fork the oracle's worker pool NOW, before any test initialises the GPU in this process (tests/oracle_pool.py:
    # the all-points comparisons of tests/test_gpu_fullsize.py).  torch.cuda.device_count() does not create a HIP context.
    expr = (config.getoption("-m", default="") or "").replace(" ", "")
    if "gpu" in expr and "notgpu" not in expr:
        try:
            import torch
            if torch.cuda.device_count() > 0:
                import oracle_pool
                oracle_pool.start()
        except Exception:       # (the tests that need the pool say so)
            pass


def pytest_unconfigure(config):
    try:
        import oracle_pool
        oracle_pool.stop()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def set_option(name, value):
    """mia_set_option for the duration of one test: the autouse fixture below restores what the test changed."""
    from torch_assimilate_amd import _cabi
    _PENDING.append((name, _cabi.set_option(name, value)))


_PENDING = []


@pytest.fixture(autouse=True)
def _restore_options():
    yield
    if _PENDING:
        from torch_assimilate_amd import _cabi
        while _PENDING:
            name, old = _PENDING.pop()
            _cabi.set_option(name, old)
