import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every context of the test-suite verifies every plan it makes (smm_plan_check; tests/test_gpu_plan_check.py)
os.environ.setdefault("SMM_CHECK", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by the product package)."""
    from oracle import oracle as _o
    _o.lib()
    return _o


@pytest.fixture(scope="session")
def ctx():
    """One GPU context for the whole session (a single process on the card)."""
    from sparse_matrix_mult_amd.engine import Context
    c = Context(0)
    yield c
    c.close()
