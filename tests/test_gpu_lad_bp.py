"""GPU parity: LAD and basis pursuit (fp64) through the C ABI vs the oracle and the README vectors."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_readme_lad_fixture(readme_lasso_xy):
    from admm_amd import admm_lad
    from oracle import entry, readme
    x, y = readme_lasso_xy
    fit = admm_lad(x, y, intercept=False).fit()
    ref = entry.admm_lad(x, y, False, entry.LAD_OPTS)
    assert fit.beta[0] == 0.0
    assert relerr(fit.beta[1:], readme.LAD_ADMM) < TOL            # README.md:139-161
    assert relerr(fit.beta, ref["beta"]) < TOL
    assert abs(fit.niter - ref["niter"]) <= max(3, 0.05 * ref["niter"])


@pytest.mark.parametrize("intercept", [True, False])
def test_lad_general_branch_vs_oracle(intercept):
    """n > 2000: the X (X'X)^-1 X' branch of ADMMLAD::next_x (ADMMLAD.h:75-76), not pinned by the README."""
    from admm_amd import admm_lad
    from oracle import entry
    rng = np.random.default_rng(21)
    n, p = 3000, 60
    x = rng.standard_normal((n, p)) * 2 + 0.3
    b = rng.uniform(size=p)
    y = x @ b + rng.standard_t(3, size=n) + 1.5
    fit = admm_lad(x, y, intercept=intercept).fit()
    ref = entry.admm_lad(x, y, intercept, entry.LAD_OPTS)
    assert relerr(fit.beta, ref["beta"]) < TOL
    assert abs(fit.niter - ref["niter"]) <= max(3, 0.05 * ref["niter"])


def test_readme_bp_fixture():
    from admm_amd import admm_bp
    from oracle import entry, readme
    x, y, bt = readme.bp_data()
    fit = admm_bp(x, y).fit()
    beta = np.asarray(fit.beta.todense()).ravel()
    e = bt - beta
    assert abs(e.min() - readme.BP_RANGE[0]) < 1e-6                # README.md:180-182
    assert abs(e.max() - readme.BP_RANGE[1]) < 1e-6
    ref = entry.admm_bp(x, y, entry.BP_OPTS)
    assert relerr(beta, ref["beta"]) < TOL
    assert abs(fit.niter - ref["niter"]) <= 2


def test_bp_perf_fixture_range():
    from admm_amd import admm_bp
    from oracle import readme
    x, y, bt = readme.bp_data(1000, 2000, 100)                    # README.md:369-393
    fit = admm_bp(x, y).fit()
    e = bt - np.asarray(fit.beta.todense()).ravel()
    assert abs(e.min() - readme.BP_PERF_RANGE[0]) < 2e-6
    assert abs(e.max() - readme.BP_PERF_RANGE[1]) < 2e-6


def test_bp_maxit_and_rho_adaptation():
    from admm_amd import admm_bp
    from oracle import entry
    rng = np.random.default_rng(4)
    n, p = 120, 400
    A = rng.standard_normal((n, p))
    bt = np.zeros(p)
    bt[rng.choice(p, 12, replace=False)] = rng.uniform(size=12)
    y = A @ bt
    for maxit in (3, 9, 10000):
        fit = admm_bp(A, y).opts(maxit=maxit).fit()
        ref = entry.admm_bp(A, y, dict(entry.BP_OPTS, maxit=maxit))
        assert abs(fit.niter - ref["niter"]) <= (0 if maxit < 100 else 2)
        assert relerr(np.asarray(fit.beta.todense()).ravel(), ref["beta"]) < TOL
