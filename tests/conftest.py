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


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a HIP device: skip the gpu-marked tests instead of failing them (the CPU suite
    is `-m "not gpu"`; on the GPU box the device is there and nothing is skipped -- there is no CPU fallback to hide)."""
    if not any("gpu" in it.keywords for it in items):
        return
    try:
        from admm_amd import _lib
        ndev = _lib.load().admm_hip_device_count()
    except Exception:
        ndev = 0
    if ndev < 1:
        skip = pytest.mark.skip(reason="no HIP device visible (libadmm_hip has no CPU fallback)")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def readme_lasso_xy():
    from oracle import readme
    return readme.lasso_data()


@pytest.fixture(autouse=True)
def _reset_library_options():
    """Variant selectors set through admm_amd.options.set(...) belong to the thread: back to the defaults after every test."""
    yield
    try:
        from admm_amd import _lib
        if _lib._lib is not None:
            _lib.options.reset()
    except Exception:      # noqa: BLE001  (library not built / no device: nothing to reset)
        pass
