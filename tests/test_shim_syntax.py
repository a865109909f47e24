"""CPU: integration/admm_shim.cpp (the Rcpp shim a maintainer of the reference adds to src/) type-checks against include/admm_hip.h.

R and Rcpp are not in the build image, so the shim has never been compiled for real (INTEGRATION.md says so).  What CAN be checked
here: with a declarations-only stand-in for the handful of Rcpp names it uses (integration/syntax_stub/Rcpp.h -- no definitions,
nothing links against it), `g++ -fsyntax-only` verifies every call into the C ABI -- argument count, order and types against the
real header -- and the shim's own C++.  The negative control shows the check bites."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CMD = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "integration", "syntax_stub"), "-I" + os.path.join(ROOT, "include")]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_shim_type_checks_against_the_c_abi(tmp_path):
    shim = os.path.join(ROOT, "integration", "admm_shim.cpp")
    r = subprocess.run(CMD + [shim], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # every RcppExport symbol the R code looks up is there
    src = open(shim).read()
    for sym in ("admm_lasso", "admm_enet", "admm_parlasso", "admm_lad", "admm_bp"):
        assert "RcppExport SEXP %s(" % sym in src, sym
    # negative control: a call with two arguments swapped must be refused
    bad = src.replace("rc = admm_hip_lasso(x.begin(), y.begin(), n, p, ADMM_MEM_HOST,", "rc = admm_hip_lasso(x.begin(), n, y.begin(), p, ADMM_MEM_HOST,", 1)
    assert bad != src
    f = tmp_path / "bad_shim.cpp"
    f.write_text(bad)
    r = subprocess.run(CMD + [str(f)], capture_output=True, text=True)
    assert r.returncode != 0
