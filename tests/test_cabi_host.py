"""CPU-side checks of the boundary: the shared library loads and exports every symbol that
include/admm_hip.h declares, the host-side mirror of the R interface validates arguments like the
R wrappers do, the Lanczos host logic matches the oracle, and without a GPU the solvers fail
loudly (no CPU fallback).  No compute calls need a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "admm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(admm_hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from admm_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 13
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert set(_lib.EXPORTS) == set(declared)
    assert b"gfx950" in lib.admm_hip_version()


def test_struct_layouts_match_header():
    from admm_amd._lib import AdmmOpts, AdmmStats
    assert ctypes.sizeof(AdmmOpts) == 32            # int + pad, 3 doubles
    assert ctypes.sizeof(AdmmStats) == 8 * 9 + 8 * 3 + 8 * 2 + 16 + 8 + 8   # 9 doubles, 3 long long, 2 doubles, 4 ints, 1 long long, factor_flops


def test_host_lanczos_matches_oracle_including_restart():
    from admm_amd import _lib
    from oracle.spectra import sym_eigs_largest
    lib = _lib.load()
    rng = np.random.default_rng(0)
    restarts = 0
    for (n, p) in [(100, 20), (400, 60), (50, 200), (1000, 150)]:
        X = (rng.standard_normal((n, p)) * 2).astype(np.float32)
        G = np.asfortranarray((X.T @ X).astype(np.float32))
        info = {}
        ev = sym_eigs_largest(lambda v: G @ v, p, 3, 10, 0.1, np.float32, info)
        out, nm = ctypes.c_float(), ctypes.c_int()
        rc = lib.admm_hip_host_lanczos(G.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), p, ctypes.byref(out), ctypes.byref(nm))
        assert rc == 0
        assert abs(out.value - float(ev)) <= 2e-6 * float(ev)
        assert nm.value == info["nmatop"]
        restarts += info["nrestart"]
        # loose under-estimate, never above lambda_max
        assert out.value <= np.linalg.eigvalsh(G.astype(np.float64))[-1] * (1 + 1e-6)
    assert restarts >= 1                              # the implicit-restart branch is exercised


def test_r_interface_argument_validation():
    from admm_amd import admm_bp, admm_enet, admm_lad, admm_lasso
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal((30, 12)), rng.standard_normal(30)
    with pytest.raises(ValueError, match="nrow\\(x\\) should be equal to length\\(y\\)"):
        admm_lasso(x, y[:-1])
    m = admm_lasso(x, y)
    assert (m.nlambda, m.lambda_min_ratio, m.maxit, m.eps_abs, m.eps_rel, m.rho, m.nthread) == (100, 1e-4, 10000, 1e-5, 1e-5, -1.0, 1)
    assert admm_lasso(x.T, y[:12]).lambda_min_ratio == 0.01                   # nrow < ncol
    with pytest.raises(ValueError, match="lambda must be positive"):
        m.penalty([0.1, -1.0])
    with pytest.raises(ValueError, match="lambda_min_ratio must be within"):
        m.penalty(nlambda=5, lambda_min_ratio=1.0)
    assert list(m.penalty([0.1, 0.5, 0.2]).lambda_) == [0.5, 0.2, 0.1]        # sorted decreasing
    with pytest.raises(ValueError, match="nthread cannot exceed ncol\\(x\\)/5"):
        m.parallel(3)
    assert m.parallel(2).nthread == 2
    with pytest.raises(ValueError, match="rho should be positive"):
        m.opts(rho=0.0)
    with pytest.raises(ValueError, match="maxit should be positive"):
        m.opts(maxit=0)
    with pytest.raises(ValueError, match="alpha must be within"):
        admm_enet(x, y).penalty(0.1, alpha=1.5)
    with pytest.raises(ValueError, match="ncol\\(x\\) must be greater than nrow\\(x\\)"):
        admm_bp(x, y)
    with pytest.raises(ValueError, match="nrow\\(x\\) must be greater than ncol\\(x\\)"):
        admm_lad(x.T, y[:12])
    b = admm_bp(x.T, y[:12])
    assert (b.maxit, b.eps_abs, b.eps_rel, b.rho) == (10000, 1e-4, 1e-4, 1.0)


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    from admm_amd import _lib
    from admm_amd._lib import AdmmOpts
    lib = _lib.load()
    x = np.asfortranarray(np.ones((4, 2)))
    y = np.ones(4)
    o = AdmmOpts(0, 1e-5, 1e-5, -1.0)
    lam = np.zeros(1)
    beta = np.zeros(3, dtype=np.float32)
    nit = np.zeros(1, dtype=np.int32)
    lo = np.zeros(1)
    rc = lib.admm_hip_lasso(x.ctypes.data, y.ctypes.data, 4, 2, 0, None, 0, 1, 1e-4, 1, 1, ctypes.byref(o),
                            lo.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                            nit.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), None)
    assert rc == 1 and b"maxit should be positive" in lib.admm_hip_last_error()
    o = AdmmOpts(10, 1e-5, 1e-5, -1.0)
    rc = lib.admm_hip_enet(x.ctypes.data, y.ctypes.data, 4, 2, 0, None, 0, 1, 1e-4, 1, 1, 1.5, ctypes.byref(o),
                           lo.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), beta.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                           nit.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), None)
    assert rc == 1 and b"alpha" in lib.admm_hip_last_error()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback():
    from admm_amd import AdmmHipError, admm_lasso
    rng = np.random.default_rng(2)
    with pytest.raises(AdmmHipError) as ei:
        admm_lasso(rng.standard_normal((40, 5)), rng.standard_normal(40)).penalty(0.1).fit()
    assert ei.value.code == 2                        # ADMM_ERR_NO_DEVICE


def test_beta_is_returned_as_csc_with_explicit_intercept_row():
    from admm_amd.api import _beta_to_csc
    dense = np.zeros((5, 2), dtype=np.float32)
    dense[2, 0] = 1.5
    dense[0, 1] = 0.25
    m = _beta_to_csc(dense)
    assert m.shape == (5, 2)
    assert list(m.indptr) == [0, 2, 3]               # row 0 stored even when it is zero (Lasso.cpp:22-30)
    assert list(m.indices) == [0, 2, 0]


def test_fit_objects_mirror_show_and_plot_data():
    """R-side conveniences of the fit classes (R/30_admm_lasso.R:181-214, R/10_admm_bp.R:111,125-133), no GPU needed."""
    import numpy as np
    from admm_amd.api import ADMM_BP, ADMM_BP_fit, ADMM_LAD_fit, ADMM_Lasso_fit
    import scipy.sparse as sp
    beta = np.zeros((4, 3), dtype=np.float32)
    beta[0] = [0.5, 0.4, 0.3]; beta[2] = [0.0, 0.1, 0.2]
    fit = ADMM_Lasso_fit(np.array([1.0, 0.5, 0.25]), beta, np.array([3, 4, 5], dtype=np.int32), {})
    assert "ADMM Lasso fitting result" in repr(fit)
    ll, coef = fit.path_data()
    assert np.allclose(ll, np.log([1.0, 0.5, 0.25])) and coef.shape == (3, 1) and np.allclose(coef[:, 0], [0.0, 0.1, 0.2])
    ax = fit.plot()
    assert ax.get_title() == "Solution path" and len(ax.lines) == 1
    one = ADMM_Lasso_fit(np.array([1.0]), beta[:, :1], np.array([3], dtype=np.int32), {})
    with pytest.raises(ValueError, match="at least two lambda"):
        one.path_data()
    assert "Basis Pursuit" in repr(ADMM_BP_fit(sp.csc_matrix(np.zeros((5, 1))), 7, {}))
    assert "LAD" in repr(ADMM_LAD_fit(np.zeros(3), 9, {}))
    m = ADMM_BP(np.zeros((3, 40)), np.zeros(3)).parallel(2)            # $parallel only stores nthread (R/10_admm_bp.R:65-76); fit -> admm_hip_parbp
    assert m.nthread == 2 and ADMM_BP(np.zeros((3, 40)), np.zeros(3)).parallel(0).nthread == 1
    # admm_dantzig is exported by the reference but calls a symbol it never builds (src/TODO/Dantzig.cpp); here the builder chain is
    # ADMM_Lasso's and fit() runs admm_hip_dantzig (GPU tests); the Lasso-only extensions refuse a Dantzig model
    import admm_amd
    dz = admm_amd.admm_dantzig(np.zeros((30, 4)), np.zeros(30)).penalty(nlambda=5).opts(maxit=10)
    assert dz._name == "ADMM Dantzig Selector model" and dz.nlambda == 5
    with pytest.raises(ValueError, match="Dantzig"):
        dz.cv(3)
    with pytest.raises(ValueError, match="Dantzig"):
        dz.fit_responses(np.zeros((30, 2)))


def test_every_option_the_library_reads_is_in_the_integration_guide_and_nothing_reads_the_environment():
    """INTEGRATION.md's option table is the only place a maintainer learns what the named options do: every name the sources pass
    to option() / option_int() must appear there.  And the sources must not call getenv at all (round 6: variant selectors belong to
    the calling thread, admm_hip_options / admm_hip_option_set; ADMM_HIP_* variables are an overlay captured once from `environ`)."""
    import glob, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names, getenvs = set(), 0
    for f in glob.glob(os.path.join(root, "admm_amd", "csrc", "*")):
        if f.endswith((".hip", ".h")):
            src = open(f).read()
            names |= set(re.findall(r'\b(?:option|option_int|env_int|env_seconds)\("([A-Z0-9_]+)"', src))
            getenvs += len(re.findall(r'\bgetenv\s*\(', src))
    assert getenvs == 0, f"{getenvs} getenv calls in admm_amd/csrc"
    guide = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in guide)
    assert len(names) > 20 and not missing, missing



def test_options_are_per_thread_and_the_typed_struct_maps_onto_the_named_ones():
    """The options ABI without a GPU: admm_hip_options_default / _set / admm_hip_option_set / _get / _reset (include/admm_hip.h)."""
    import ctypes
    import threading
    from admm_amd import _lib
    lib = _lib.load()
    _lib.options.reset()
    assert lib.admm_hip_option_get(b"GRAM_SPLIT") is None
    o = _lib.AdmmHipOptions()
    _lib.check(lib.admm_hip_options_default(ctypes.byref(o)))
    assert o.struct_size == ctypes.sizeof(_lib.AdmmHipOptions) and o.gram_split == 0
    o.gram_split, o.inverse_precision, o.consensus_two_pass, o.batch_iters = 3, 2, 1, 32
    _lib.check(lib.admm_hip_options_set(ctypes.byref(o)))
    assert lib.admm_hip_option_get(b"GRAM_SPLIT") == b"bf16x3" and lib.admm_hip_option_get(b"INVERSE") == b"f64"
    assert lib.admm_hip_option_get(b"ADMM_HIP_PAR_ONEPASS") == b"0" and lib.admm_hip_option_get(b"batch_iters") == b"32"
    seen = {}
    t = threading.Thread(target=lambda: seen.update(other=lib.admm_hip_option_get(b"GRAM_SPLIT")))
    t.start(); t.join()
    assert seen["other"] is None                                   # another thread sees the defaults
    with _lib.options(GRAM_SPLIT="f16x2", WIDE_PERSIST=0):
        assert lib.admm_hip_option_get(b"GRAM_SPLIT") == b"f16x2" and lib.admm_hip_option_get(b"WIDE_PERSIST") == b"0"
    assert lib.admm_hip_option_get(b"GRAM_SPLIT") == b"bf16x3" and lib.admm_hip_option_get(b"WIDE_PERSIST") is None
    o.gram_split = 7
    assert lib.admm_hip_options_set(ctypes.byref(o)) != 0          # rejected, with a message
    assert b"gram_split" in lib.admm_hip_last_error()
    _lib.check(lib.admm_hip_options_set(None))
    assert lib.admm_hip_option_get(b"INVERSE") is None


def test_bench_refuses_a_world_size_that_disagrees_with_gpus_without_a_gpu():
    """`python bench.py --gpus 8` under WORLD_SIZE=1 must refuse before it touches a device (round-5 review: it ran one GPU and printed
    n_gpus: 1); the check sits in front of every import that needs one, so it is testable here."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout), (r.returncode, r.stderr[-300:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
