"""GPU: every solver family once through the host-sanitized build of the library (ASan + UBSan on the host half of every
translation unit; tests/test_sanitizers.py has the GPU-free half): plan construction and teardown, the batch-enqueue loop
drivers, trace / iterate-dump read-back, result marshalling, an error path that unwinds through device buffers."""
import pytest

from test_sanitizers import run_child

pytestmark = pytest.mark.gpu


def test_solver_families_under_host_asan_and_ubsan():
    out = run_child("gpu_paths.py", timeout=900)
    assert "gpu paths ok" in out
