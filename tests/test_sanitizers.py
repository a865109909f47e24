"""CPU: the HOST half of libadmm_hip under AddressSanitizer + UndefinedBehaviorSanitizer (admm_amd/build.py build_sanitized:
every translation unit's host code with -fsanitize=address,undefined, the device code as usual) on the paths that need no
GPU -- argument validation and error strings, struct round trips, the host Lanczos driver, the no-device path.  The GPU half
(plan construction, loop drivers, marshalling) is tests/test_gpu_sanitizers.py.  SURVEY.md section 5 (sanitizers)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run_child(script, timeout=600, expect_report=False):
    from admm_amd import build as B
    rt = B.sanitizer_runtime()
    if rt is None or not os.path.exists(B.LIB_SAN):
        pytest.skip("no sanitizer runtime / libadmm_hip_san.so not built (python -c 'from admm_amd import build; build.build_sanitized()')")
    if not B._lib_is_current(B.LIB_SAN):           # linked from other sources than those beside it (__graft_entry__.build() keeps it current)
        B.build_sanitized(verbose=False)
    env = dict(os.environ, LD_PRELOAD=rt, ADMM_HIP_LIB=B.LIB_SAN,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0:detect_odr_violation=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "sanitizer", script)], env=env, capture_output=True, text=True, timeout=timeout)
    out = r.stdout + r.stderr
    if expect_report:
        return r.returncode, out
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    return out


def test_host_only_paths_under_asan_and_ubsan():
    out = run_child("host_paths.py")
    assert "host paths ok" in out


def test_negative_control_the_sanitized_build_reports_a_planted_overflow():
    rc, out = run_child("negative_control.py", expect_report=True)
    assert rc != 0 and "AddressSanitizer: heap-buffer-overflow" in out and "not reported" not in out, out[-2000:]
