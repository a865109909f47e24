import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


try:
    import torch  # noqa: F401  -- before libadmm_hip so both share ONE HIP runtime (same SONAMEs)
except Exception:  # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def readme_lasso_xy():
    from oracle import readme
    return readme.lasso_data()
