"""The compiled C restatement of the tall loop (oracle/c) against the README known-answer vectors and the NumPy oracle."""
import numpy as np

from helpers import relerr, synth_lasso


def test_c_oracle_readme_lasso_and_enet(readme_lasso_xy):
    from oracle import ctall, entry, readme
    x, y = readme_lasso_xy
    r = ctall.admm_lasso_c(x, y, [readme.LAMBDA], 100, 1e-4, True, True, entry.LASSO_OPTS)
    assert relerr(r["beta"][:, 0], readme.LASSO_ADMM) < 1e-4          # README.md:66-88 admm column
    ref = entry.admm_lasso(x, y, [readme.LAMBDA], 100, 1e-4, True, True, entry.LASSO_OPTS)
    assert int(r["niter"][0]) == int(ref["niter"][0]) == 31
    assert abs(r["rho"] - 13.678) < 0.01
    r = ctall.admm_enet_c(x, y, [readme.LAMBDA], 100, 1e-4, True, True, 0.5, entry.LASSO_OPTS)
    assert relerr(r["beta"][:, 0], readme.ENET_ADMM) < 1e-4           # README.md:100-123
    assert int(r["niter"][0]) == 22


def test_c_oracle_path_vs_numpy_oracle():
    from oracle import ctall, entry
    x, y = synth_lasso(600, 80, 8, seed=2)
    ref = entry.admm_lasso(x, y, None, 12, 1e-4, True, True, entry.LASSO_OPTS)
    for mode, nt in ((0, 1), (1, 2)):
        r = ctall.admm_lasso_c(x, y, None, 12, 1e-4, True, True, entry.LASSO_OPTS, mode=mode, nthreads=nt)
        assert np.allclose(r["lambda"], ref["lambda"])
        # the first half of the path is unaffected by stopping-rule flips
        assert np.abs(r["niter"][:6].astype(int) - ref["niter"][:6].astype(int)).max() <= 1, (r["niter"], ref["niter"])
        for j in range(12):
            assert relerr(r["beta"][:, j], ref["beta"][:, j]) < 2e-3, (mode, j)
        assert relerr(r["beta"][:, :6], ref["beta"][:, :6]) < 1e-4
    opts = dict(entry.LASSO_OPTS, maxit=4)
    r = ctall.admm_lasso_c(x, y, [0.3, 0.05], 100, 1e-4, True, True, opts)
    ref = entry.admm_lasso(x, y, [0.3, 0.05], 100, 1e-4, True, True, opts)
    assert list(r["niter"]) == list(ref["niter"]) == [5, 5]
    assert relerr(r["beta"], ref["beta"]) < 1e-5


def test_c_oracle_clean_under_asan_and_ubsan(tmp_path):
    """The C restatement compiled with -fsanitize=address,undefined and driven by a small C main on a toy problem
    (X'X + rho I factorised here in Python, handed over as text): no report, exit code 0.  (the
    reference has no sanitizer runs; its latent UB is listed there.)"""
    import os
    import subprocess
    import scipy.linalg as sla
    from helpers import synth_lasso
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "oracle", "c", "admm_tall_cpu.c")
    x, y = synth_lasso(200, 24, 4, seed=3)
    xs = ((x - x.mean(0)) / x.std(0)).astype(np.float32)
    ys = ((y - y.mean()) / y.std()).astype(np.float32)
    p = xs.shape[1]
    G = (xs.T @ xs).astype(np.float32)
    rho = 30.0
    L = np.tril(sla.cho_factor(G + rho * np.eye(p, dtype=np.float32), lower=True)[0]).astype(np.float32)
    XY = (xs.T @ ys).astype(np.float32)
    main_c = tmp_path / "main.c"

    def arr(a):
        return ", ".join(repr(float(v)) + "f" for v in np.asfortranarray(a).ravel(order="F"))

    lam0 = float(np.abs(XY).max())
    template = r"""
#include <stdio.h>
int oracle_tall_path(const float*, const float*, int, const double*, int, double, double, double, int, double, int, int,
                     float*, int*, double*);
static const float L[] = {@L@};
static const float XY[] = {@XY@};
int main(void) {
    double lam[3] = {@LAM0@, @LAM1@, @LAM2@};
    float beta[3 * @P@]; int niter[3]; double secs;
    for (int mode = 0; mode < 2; ++mode) {
        int rc = oracle_tall_path(L, XY, @P@, lam, 3, @RHO@, 1e-5, 1e-5, 500, mode ? 0.5 : -1.0, 0, 2, beta, niter, &secs);
        if (rc) return 1;
    }
    printf("%d %d %d\n", niter[0], niter[1], niter[2]);
    return 0;
}
"""
    code = (template.replace("@L@", arr(L)).replace("@XY@", arr(XY)).replace("@P@", str(p)).replace("@RHO@", repr(rho))
            .replace("@LAM0@", repr(lam0)).replace("@LAM1@", repr(0.3 * lam0)).replace("@LAM2@", repr(0.05 * lam0)))
    main_c.write_text(code)
    exe = tmp_path / "oracle_asan"
    r = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fopenmp", str(main_c), src,
                        "-o", str(exe), "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (r.stdout, r.stderr[-2000:])
    assert len(r.stdout.split()) == 3


def test_c_oracle_decision_trace_matches_numpy_oracle():
    """The decision trace the compiled loop records (used by the p = 10 000 fixture, tests/golden/make_c2_short.py): same
    (lambda, iteration) sequence, same outcomes and the same residuals / thresholds as the NumPy oracle's own trace on the
    first lambdas of a path (before any rounding-level count flip), and consistent with the iteration counts returned."""
    from oracle import ctall, entry
    x, y = synth_lasso(600, 80, 8, seed=2)
    tr = []
    r = ctall.admm_lasso_c(x, y, None, 6, 1e-4, True, True, entry.LASSO_OPTS, mode=0, nthreads=1, trace=tr)
    tr = np.asarray(tr)
    d = {"trace": []}
    ref = entry.admm_lasso(x, y, None, 6, 1e-4, True, True, entry.LASSO_OPTS, d)
    rt = np.asarray(d["trace"], dtype=np.float64)
    assert len(tr) == int(np.sum(r["niter"])) and len(rt) == int(np.sum(ref["niter"]))
    # the two restatements round their triangular solves differently (own loops vs LAPACK), so somewhere along the path a
    # near-tie flips an iteration count (tests/test_flip_floor.py): compare the lambdas before the first such flip
    same = 0
    while same < 6 and r["niter"][same] == ref["niter"][same]:
        same += 1
    assert same >= 3, (r["niter"], ref["niter"])
    a, b = tr[tr[:, 0] < same], rt[rt[:, 0] < same]
    assert np.array_equal(a[:, 0], b[:, 0]) and np.array_equal(a[:, 1], b[:, 1])
    assert np.array_equal(a[:, 7].astype(int), b[:, 8].astype(int))                    # outcome: 0 stop, 1 accelerate, 2 restart
    assert np.allclose(a[:, 2], b[:, 2], rtol=1e-4) and np.allclose(a[:, 3], b[:, 3], rtol=1e-4)      # eps_p, eps_d
    assert np.allclose(a[:, 4], b[:, 4], rtol=1e-2)                                    # r_p
    big = b[:, 5] > b[:, 3]                      # r_d = rho ||z - z_old||: single-ulp flips of a few z entries once it is below eps_d
    assert big.sum() > 20 and np.allclose(a[big, 5], b[big, 5], rtol=0.1)
    for l in range(6):                                                                 # one "stop" record ends every converged lambda
        rows = tr[tr[:, 0] == l]
        assert len(rows) == r["niter"][l] and rows[-1, 7] == 0 and (rows[:-1, 7] > 0).all()


def test_c2_short_fixture_is_self_consistent():
    """tests/golden/c2_short_path.npz (the compiled oracle at p = 10 000; the GPU test that uses it is
    tests/test_gpu_fullsize.py::test_c2_width_short_path_vs_compiled_oracle_fixture): every recorded decision is the rule
    applied to the residuals recorded with it, the counts are the trace's, and both kinds of exit occur."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_short_path.npz"))
    t, niter, maxit = g["trace"], g["niter"], int(g["maxit"])
    conv = (t[:, 4] < t[:, 2]) & (t[:, 5] < t[:, 3])
    assert np.array_equal(conv, t[:, 7] == 0)
    for l in range(len(niter)):
        rows = t[t[:, 0] == l]
        assert len(rows) == min(int(niter[l]), maxit)
        assert (rows[-1, 7] == 0) == (niter[l] <= maxit)
    c, code = t[:, 6], t[:, 7]
    c_old = 9999.0
    for k in range(len(t)):                         # the restart rule on the recorded c (FADMMBase.h:243-256), chained across lambdas
        if code[k] == 0:
            continue
        assert (c[k] < 0.999 * c_old) == (code[k] == 1), k
        c_old = c[k] if code[k] == 1 else c_old / 0.999
    assert (niter == maxit + 1).any() and (niter <= maxit).any()
    assert g["beta"].shape == (int(g["p"]) + 1, len(niter)) and np.isfinite(g["beta"]).all()
