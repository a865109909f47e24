"""GPU parity: tall Lasso / Elastic-net through the C ABI vs the CPU oracle (and the README vectors)."""
import numpy as np
import pytest

from helpers import assert_tall_parity, relerr, synth_lasso, traced_fit, traced_parity

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: beta within 1e-4 relative (norm-wise)


def _problem(x, y, nl, standardize=True, intercept=True, alpha=None, lam=None, opts=None):
    from oracle import entry
    return dict(x=x, y=y, lam=lam, nlambda=nl, lmin_ratio=1e-4, standardize=standardize, intercept=intercept,
                opts=opts or entry.LASSO_OPTS, alpha=alpha)


def test_readme_lasso_fixture(readme_lasso_xy):
    from admm_amd import admm_lasso
    from oracle import entry, readme
    x, y = readme_lasso_xy
    # on the decision trace: iteration count IDENTICAL to the oracle's (31), beta within 1e-4 of it
    fit, rep = traced_parity(admm_lasso(x, y).penalty(readme.LAMBDA), _problem(x, y, 100, lam=[readme.LAMBDA]), TOL, label="README lasso")
    beta = fit.beta_dense[:, 0]
    assert relerr(beta, readme.LASSO_ADMM) < TOL                 # README.md:66-88 admm column
    assert np.array_equal(beta != 0, readme.LASSO_ADMM != 0)
    assert int(fit.niter[0]) == 31 and len(rep["forced"]) == 0
    assert abs(fit.stats["rho"] - 13.678) < 0.01
    plain = admm_lasso(x, y).penalty(readme.LAMBDA).fit()         # the plain entry point is the same execution
    assert np.array_equal(plain.beta_dense, fit.beta_dense) and list(plain.niter) == list(fit.niter)


def test_readme_enet_fixture(readme_lasso_xy):
    from admm_amd import admm_enet
    from oracle import entry, readme
    x, y = readme_lasso_xy
    fit, rep = traced_parity(admm_enet(x, y).penalty(readme.LAMBDA, alpha=0.5), _problem(x, y, 100, alpha=0.5, lam=[readme.LAMBDA]), TOL,
                             label="README enet")
    beta = fit.beta_dense[:, 0]
    assert relerr(beta, readme.ENET_ADMM) < TOL                  # README.md:100-123
    assert np.array_equal(beta != 0, readme.ENET_ADMM != 0)
    assert int(fit.niter[0]) == 22 and len(rep["forced"]) == 0   # 22 iterations


@pytest.mark.parametrize("standardize,intercept", [(True, True), (True, False), (False, True), (False, False)])
def test_tall_path_vs_oracle(standardize, intercept):
    """20-lambda warm-started path for every DataStd flag, judged on the decision trace (helpers.assert_tall_parity):
    the oracle follows the GPU through near-tie threshold tests only; iteration counts identical and every column
    within 1e-4 on that common trajectory."""
    from admm_amd import admm_lasso
    x, y = synth_lasso(2000, 300, 30, seed=7)
    x += 0.7                                                     # non-zero column means so the flags matter
    fit, trace = traced_fit(admm_lasso(x, y, intercept=intercept, standardize=standardize).penalty(nlambda=20))
    rep = assert_tall_parity(fit.beta_dense, fit.niter, trace, _problem(x, y, 20, standardize, intercept), TOL,
                             label=f"std={int(standardize)} icpt={int(intercept)}")
    assert np.allclose(fit.lambda_, rep["ref"]["lambda"], rtol=1e-5)
    # (the ceiling on decisions taken from the GPU -- helpers R4 -- is asserted inside assert_tall_parity)
    assert len(rep["loose"]) == 0, rep["loose"]
    # the plain entry point gives the same result as the prepared-problem one the trace came from
    fit2 = admm_lasso(x, y, intercept=intercept, standardize=standardize).penalty(nlambda=20).fit()
    assert np.array_equal(fit2.beta_dense, fit.beta_dense) and list(fit2.niter) == list(fit.niter)
    # first lambda = lambda_max: all coefficients zero
    assert np.count_nonzero(fit.beta_dense[1:, 0]) == 0


def test_tall_enet_path_vs_oracle():
    from admm_amd import admm_enet
    x, y = synth_lasso(1500, 200, 20, seed=11)
    fit, trace = traced_fit(admm_enet(x, y).penalty(nlambda=15, alpha=0.6))
    rep = assert_tall_parity(fit.beta_dense, fit.niter, trace, _problem(x, y, 15, alpha=0.6), TOL, label="enet")
    assert len(rep["loose"]) == 0, rep["loose"]


@pytest.mark.parametrize("p,n", [(2048, 5000), (2300, 4700)])
def test_tall_symmetric_xupdate_path_vs_oracle(p, n):
    """p >= 2048: the lower-triangle symmetric mat-vec (the headline kernel) inside the solver, against the oracle."""
    from admm_amd import admm_lasso
    x, y = synth_lasso(n, p, 40, seed=p)
    fit, trace = traced_fit(admm_lasso(x, y).penalty(nlambda=8))
    assert fit.stats["xupdate_variant"] == 1
    rep = assert_tall_parity(fit.beta_dense, fit.niter, trace, _problem(x, y, 8), TOL, label=f"sym p={p}")
    assert len(rep["loose"]) == 0, rep["loose"]


def test_tall_ragged_and_maxit():
    """p not a multiple of anything, user lambda grid, and the maxit exit (niter = maxit + 1)."""
    from admm_amd import admm_lasso
    from oracle import entry
    x, y = synth_lasso(523, 97, 9, seed=3)
    lam = [0.5, 0.1, 0.02]
    opts = dict(entry.LASSO_OPTS, maxit=5)
    fit, _ = traced_parity(admm_lasso(x, y).penalty(lam).opts(maxit=5), _problem(x, y, 100, lam=lam, opts=opts), TOL, label="ragged maxit 5")
    assert list(fit.niter) == [6, 6, 6]
    fit, _ = traced_parity(admm_lasso(x, y).penalty(lam), _problem(x, y, 100, lam=lam), TOL, label="ragged")
    assert fit.niter.max() < 10000


def test_device_resident_input_matches_host_input():
    """ADMM_MEM_DEVICE path (what bench.py uses) gives the same result as host input."""
    import torch
    from admm_amd import admm_lasso, DevicePtr
    x, y = synth_lasso(1200, 150, 15, seed=5)
    fit_h = admm_lasso(x, y).penalty(nlambda=8).fit()
    xd = torch.tensor(np.asfortranarray(x).T.copy(), device="cuda")      # p x n row-major == n x p column-major
    yd = torch.tensor(y, device="cuda")
    torch.cuda.synchronize()
    fit_d = admm_lasso(DevicePtr(xd.data_ptr()), DevicePtr(yd.data_ptr()), n=1200, p=150).penalty(nlambda=8).fit()
    assert np.array_equal(fit_h.beta_dense, fit_d.beta_dense)
    assert list(fit_h.niter) == list(fit_d.niter)
