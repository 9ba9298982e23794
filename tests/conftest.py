import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger-size GPU property tests")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_norm(a, b):
    """Array-normalised error: max|a-b| / max|b| (the metric of rounds 1-3)."""
    import numpy as np
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def elementwise_excess(a, b, rtol=1e-5, floor=1e-9):
    """Element-wise criterion asked for by VERDICT r3 (weak item 2): every entry must satisfy
    |a - b| <= rtol*|b| + floor*max|b| (north-star: "within 1e-5 relative"); returns the largest ratio
    |a-b| / (rtol*|b| + floor*max|b|) -- <= 1 passes -- so a failure message says by how much."""
    import numpy as np
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    if a.size == 0:
        return 0.0
    bound = rtol * np.abs(b) + floor * (np.max(np.abs(b)) + 1e-300)
    return float(np.max(np.abs(a - b) / bound))


def assert_parity(a, b, what="", norm_tol=1e-8, rtol=1e-5, floor=1e-9):
    """Both yardsticks at once: the array-normalised 1e-8 AND the element-wise 1e-5 relative criterion."""
    rn = rel_norm(a, b)
    assert rn < norm_tol, (what, "norm", rn)
    ex = elementwise_excess(a, b, rtol, floor)
    assert ex <= 1.0, (what, "element-wise excess", ex)
