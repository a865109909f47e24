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
