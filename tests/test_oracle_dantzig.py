"""CPU: the oracle's restatement of the reference's UNBUILT Dantzig selector (src/TODO/ADMMDantzig.h, Dantzig.cpp) has no
vector of the reference to be pinned on -- `admm_dantzig()` fails in R with a missing symbol -- so it is pinned on the
problem itself: min ||beta||_1 s.t. ||X'(X beta - y)||_inf <= lambda is a linear programme, solved here with SciPy's HiGHS."""
import numpy as np
import pytest

from helpers import synth_lasso


def _lp_dantzig(xs, ys, lam):
    """beta = u - v, u, v >= 0:  min 1'(u + v)  s.t.  -lam <= X'X (u - v) - X'y <= lam."""
    from scipy.optimize import linprog
    p = xs.shape[1]
    G = xs.T @ xs
    c = xs.T @ ys
    A = np.block([[G, -G], [-G, G]])
    b = np.concatenate([lam + c, lam - c])
    res = linprog(np.ones(2 * p), A_ub=A, b_ub=b, bounds=[(0, None)] * (2 * p), method="highs")
    assert res.status == 0, res.message
    return res.x[:p] - res.x[p:], res.fun


@pytest.mark.parametrize("n,p", [(500, 100), (400, 60)])
def test_dantzig_oracle_solves_the_linear_programme(n, p):
    from oracle import entry
    x, y = synth_lasso(n, p, 4, seed=71)
    d = {"trace": []}
    opts = dict(maxit=20000, eps_abs=1e-6, eps_rel=1e-6, rho=-1.0)
    ref = entry.admm_dantzig(x, y, None, 5, 0.05, True, True, opts, d)
    std, sol = d["std"], d["solver"]
    print("niter", ref["niter"])
    assert np.count_nonzero(ref["beta"][1:, 0]) == 0                      # lambda_max = max|X'y|: the null model
    xs = (x - x.mean(0)) / x.std(0)
    ys = (y - y.mean()) / y.std()
    for j in (2, 4):
        lam_int = ref["lambda"][j] * n / float(std.scaleY)
        b_std = ref["beta"][1:, j] * np.asarray(std.scaleX) / float(std.scaleY)           # back to the solver's units
        viol = np.abs(xs.T @ (xs @ b_std - ys)).max() - lam_int
        _, fstar = _lp_dantzig(xs, ys, lam_int)
        assert ref["niter"][j] <= 20000, (j, ref["niter"])
        assert viol < 1e-3 * lam_int, (j, viol, lam_int)                   # feasible to the solver's tolerance
        assert abs(np.abs(b_std).sum() - fstar) < 2e-3 * max(fstar, 1e-3), (j, np.abs(b_std).sum(), fstar)     # and optimal
    assert sol.rho > 0 and sol.sprad > 0


def test_the_unbuilt_dantzig_algorithm_does_not_converge_for_p_greater_than_n():
    """Why `admm_dantzig` stays where the reference leaves it: the algorithm of src/TODO/ADMMDantzig.h,
    run under the CURRENT ADMMBase::solve with its rho adaptation, stalls on most problems that are not comfortably tall --
    residuals hover a factor 2-5 above their thresholds while rho is multiplied and divided every iteration (the step
    1 / gamma uses the SQUARE of the loose Lanczos value, 6-16 % below ||X'X||^2, so the linearisation does not majorise).
    Pinned here so that the claim is a test, not prose: p > n never converges within the R default maxit = 10 000, at any of
    the tolerances 1e-4 .. 1e-6, from the second lambda on; a tall problem with n = 6.7 p loses lambdas too."""
    from oracle import entry
    x, y = synth_lasso(60, 90, 4, seed=71)
    for eps in (1e-4, 1e-6):
        ref = entry.admm_dantzig(x, y, None, 5, 0.05, True, True, dict(maxit=10000, eps_abs=eps, eps_rel=eps, rho=-1.0))
        assert ref["niter"][0] == 2 and np.all(ref["niter"][1:] == 10001), (eps, ref["niter"])
    x, y = synth_lasso(200, 30, 4, seed=71)
    ref = entry.admm_dantzig(x, y, None, 8, 0.05, True, True, dict(maxit=10000, eps_abs=1e-5, eps_rel=1e-5, rho=-1.0))
    assert (ref["niter"] == 10001).sum() >= 3, ref["niter"]
