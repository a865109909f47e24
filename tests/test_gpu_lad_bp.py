"""GPU parity: LAD and basis pursuit (fp64) through the C ABI vs the oracle and the README vectors."""
import numpy as np
import pytest

from helpers import assert_dense_followed, dense_state_records, dense_stepwise, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_readme_lad_fixture(readme_lasso_xy):
    from admm_amd import admm_lad
    from oracle import entry, readme
    x, y = readme_lasso_xy
    fit = admm_lad(x, y, intercept=False).fit(trace=True, state=dense_state_records(len(y), 10000))
    assert fit.beta[0] == 0.0
    assert relerr(fit.beta[1:], readme.LAD_ADMM) < TOL            # README.md:139-161
    assert_dense_followed("lad", fit.beta, fit.niter, fit.trace, x, y, entry.LAD_OPTS, intercept=False, tol=1e-8, label="README LAD (hat-matrix branch)")
    dense_stepwise("lad", fit, x, y, entry.LAD_OPTS, intercept=False, label="README LAD (hat-matrix branch)")


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
    fit = admm_lad(x, y, intercept=intercept).fit(trace=True, state=dense_state_records(n, 10000))
    assert_dense_followed("lad", fit.beta, fit.niter, fit.trace, x, y, entry.LAD_OPTS, intercept=intercept, tol=1e-8, label=f"LAD n=3000 icpt={int(intercept)}")
    dense_stepwise("lad", fit, x, y, entry.LAD_OPTS, intercept=intercept, label=f"LAD n=3000 icpt={int(intercept)}")



@pytest.mark.parametrize("n,p,maxit,onepass", [(3001, 60, 10000, "1"), (2600, 1100, 60, "1"), (4100, 2300, 30, "1"), (4300, 3300, 25, "1"), (5203, 4200, 25, "1"),
                                                 (6200, 5300, 20, "1"), (12000, 6200, 12, "1"), (3001, 60, 10000, "0"), (2600, 1100, 60, "0"), (4100, 2300, 30, "0")])
def test_lad_one_pass_and_two_pass_forms_vs_oracle(n, p, maxit, onepass):
    """The general branch in both forms: round 6's one pass over the ROWS of X per iteration (lad_rows_kernel: x = X s, the prox, the dual
    and X'z_new, X'y_new from the same rows; X'vec from the p-vectors X'd, X'z, X'y) and the reference's two products (LAD_ONEPASS=0).
    Every register layout of the rows kernel (1 ... 6 double2 per thread and row; 6200 columns are beyond it: two-pass either way), ragged
    row runs; each form held to the oracle by the follow rule at 1e-8 and by the stepwise instrument (adj, z, y bit for bit; the projection
    against Householder QR)."""
    from admm_amd import admm_lad, options
    from oracle import entry
    rng = np.random.default_rng(100 + p)
    x = rng.standard_normal((n, p)) * 2 + 0.3
    b = rng.uniform(size=p) / np.sqrt(p)
    y = x @ b + rng.standard_t(3, size=n) + 1.5
    opts = dict(entry.LAD_OPTS, maxit=maxit)
    with options(LAD_ONEPASS=onepass):
        fit = admm_lad(x, y, intercept=True).opts(maxit=maxit).fit(trace=True, state=dense_state_records(n, maxit))
    assert fit.stats["xupdate_variant"] == (1 if onepass == "1" and p <= 6144 else 0)
    label = f"LAD n={n} p={p} onepass={onepass}"
    assert_dense_followed("lad", fit.beta, fit.niter, fit.trace, x, y, opts, intercept=True, tol=1e-8, label=label)
    dense_stepwise("lad", fit, x, y, opts, intercept=True, label=label)


def test_readme_bp_fixture():
    from admm_amd import admm_bp
    from oracle import entry, readme
    x, y, bt = readme.bp_data()
    fit = admm_bp(x, y).fit(trace=True, state=dense_state_records(x.shape[1], 10000))
    beta = np.asarray(fit.beta.todense()).ravel()
    e = bt - beta
    assert abs(e.min() - readme.BP_RANGE[0]) < 1e-6                # README.md:180-182
    assert abs(e.max() - readme.BP_RANGE[1]) < 1e-6
    assert_dense_followed("bp", beta, fit.niter, fit.trace, x, y, entry.BP_OPTS, tol=1e-8, label="README BP")
    dense_stepwise("bp", fit, x, y, entry.BP_OPTS, label="README BP")


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
        fit = admm_bp(A, y).opts(maxit=maxit).fit(trace=True, state=dense_state_records(p, maxit))
        assert_dense_followed("bp", np.asarray(fit.beta.todense()).ravel(), fit.niter, fit.trace, A, y, dict(entry.BP_OPTS, maxit=maxit),
                              tol=1e-8, label=f"BP maxit={maxit}")
        dense_stepwise("bp", fit, A, y, dict(entry.BP_OPTS, maxit=maxit), label=f"BP maxit={maxit}")
