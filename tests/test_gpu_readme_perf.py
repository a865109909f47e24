"""GPU: the problems of the reference README's PERFORMANCE section (/root/reference/README.md:195-419) through the C ABI.

tests/test_oracle_readme_perf.py pins the ORACLE on the numbers the README prints for them (the consensus rows to 3-4 digits,
LAD / BP to all printed digits, the serial Lasso rows sandwiched between the exact optimum and an execution with glmnet's
stopping rule).  Here the library runs the same problems: it must reproduce the same printed numbers where the oracle does, and
on every problem it is held to the oracle by the trace rule (counts identical, every column 1e-4) -- the wide solver also by the
stepwise rule on its iterate dump (helpers.traced_parity)."""
import numpy as np
import pytest

import readme_perf_cases as R
from helpers import traced_parity

pytestmark = [pytest.mark.gpu, pytest.mark.filterwarnings("ignore")]
TOL = 1e-4


def _range(a, b):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(d.min()), float(d.max())


@pytest.mark.parametrize("shape,kind", [("wide", "lasso"), ("wide", "enet"), ("wide", "padmm"), ("tall", "lasso"), ("tall", "enet"), ("tall", "padmm")])
def test_readme_lasso_family_on_glmnets_grid(shape, kind):
    """README.md:211-289: admm_lasso(x, y)$penalty(lambdas1)$fit(), ...$parallel()$fit(), admm_enet(x, y)$penalty(lambdas2, alpha = 0.6)$fit()."""
    from admm_amd import admm_enet, admm_lasso
    from oracle import entry
    n, p = (10000, 1000) if shape == "tall" else (1000, 2000)
    x, y = R.lasso_data(n, p)
    alpha = R.ENET_ALPHA if kind == "enet" else 1.0
    lam, exact = R.glmnet_grid_and_optimum(n, p, alpha)
    prob = dict(x=x, y=y, lam=lam, nlambda=100, lmin_ratio=1e-4, standardize=True, intercept=True, opts=entry.LASSO_OPTS,
                alpha=alpha if kind == "enet" else None)
    if kind == "enet":
        model = admm_enet(x, y).penalty(lam, alpha=alpha)
    else:
        model = admm_lasso(x, y).penalty(lam)
        if kind == "padmm":
            model = model.parallel()                             # default nthread = 2 (R/30_admm_lasso.R:95)
            prob["nthread"] = 2
    fit, rep = traced_parity(model, prob, TOL, label=f"README perf {shape} {kind}")
    assert fit.stats["branch"] == (2 if kind == "padmm" else (0 if shape == "tall" else 1))
    lo, hi = _range(exact, fit.beta_dense)
    olo, ohi = _range(exact, rep["ref"]["beta"])
    readme = (R.README_TALL if shape == "tall" else R.README_WIDE)[kind]
    print(f"[README perf {shape} {kind}] exact - libadmm_hip {lo:.6e} {hi:.6e} | exact - oracle {olo:.6e} {ohi:.6e} | README (glmnet - admm) {readme}")
    assert abs(lo - olo) < 2e-5 and abs(hi - ohi) < 2e-5
    if kind == "padmm":                                          # this package's own error dominates: the printed digits
        assert abs(lo - readme[0]) <= 1e-2 * abs(readme[0]), (lo, readme)
    for e, rd in zip((lo, hi), readme):
        assert abs(e) <= 1.6 * abs(rd) + 1e-6, (shape, kind, (lo, hi), readme)


def test_readme_lad_cases_against_the_linear_programme():
    """README.md:299-364: rq.fit - admm_lad(x, y, intercept = FALSE): n = 1000 (hat-matrix branch) all printed digits against the
    LP's vertex, n = 5000 (general branch) two digits against the LP optimum (fixture)."""
    from admm_amd import admm_lad
    x, y = R.lad_data(1000, 500)
    beta_lp, _ = R.lad_lp(x, y)
    fit = admm_lad(x, y, intercept=False).fit()
    lo, hi = _range(beta_lp, fit.beta[1:])
    print(f"[README LAD n=1000] LP - libadmm_hip {lo:.9f} {hi:.9f} | README {R.README_LAD_1000}; {fit.niter} iterations")
    assert abs(lo - R.README_LAD_1000[0]) < 5e-9 and abs(hi - R.README_LAD_1000[1]) < 5e-9
    from helpers import assert_dense_followed, dense_state_records, dense_stepwise
    from oracle import entry
    x, y = R.lad_data(5000, 1000)
    beta_lp, _ = R.lad_lp_n5000()
    fit = admm_lad(x, y, intercept=False).fit(trace=True, state=dense_state_records(5000, 2000))
    lo, hi = _range(beta_lp, fit.beta[1:])
    print(f"[README LAD n=5000] LP - libadmm_hip {lo:.7f} {hi:.7f} | README {R.README_LAD_5000}; {fit.niter} iterations")
    # every iteration the reference's (stepwise), the whole run the oracle's once it takes the library's side of near-ties (follow rule)
    dense_stepwise("lad", fit, x, y, entry.LAD_OPTS, intercept=False, label="README LAD n=5000 (general branch)")
    assert_dense_followed("lad", fit.beta, fit.niter, fit.trace, x, y, entry.LAD_OPTS, intercept=False, tol=1e-8, label="README LAD n=5000 (general branch)")
    # The printed range to ONE digit only, for a stated reason.  This problem has an exact tie at iteration 2: while z = 0 the
    # iterate repeats (x = P(2 y - x_0) = x_0 in exact arithmetic), and the restart test compares c_2 = c_1 with
    # 0.999 (c_0 / 0.999) = c_0 (1 +- ulp) -- the order in which ||r||^2 is added up decides.  NumPy's order restarts (248
    # iterations, range -0.0035811 0.0041068: the README's two digits, tests/test_oracle_readme_perf.py); this library's order
    # accelerates (328 iterations) and ends, a different momentum schedule later, at another point inside the same eps = 1e-4
    # ball around the LP optimum.  Which side Eigen's order takes is not knowable here; both are executions of FADMMBase.h:243.
    assert abs(lo - R.README_LAD_5000[0]) < 4e-4 and abs(hi - R.README_LAD_5000[1]) < 4e-4
    _, obj = R.lad_lp_n5000()
    assert obj <= np.abs(y - x @ fit.beta[1:]).sum() <= obj * (1 + 5e-3)


def test_readme_bp_p10000():
    """README.md:397-419: n = 1000, p = 10000, nsig = 200: range(beta_true - out_admm$beta), all printed digits."""
    from admm_amd import admm_bp
    from oracle import readme
    x, y, bt = readme.bp_data(1000, 10000, 200)
    fit = admm_bp(x, y).fit()
    lo, hi = _range(bt, fit.beta.toarray().ravel())
    print(f"[README BP p=10000] truth - libadmm_hip {lo:.7f} {hi:.7f} | README {R.README_BP_10000}; {fit.niter} iterations")
    assert abs(lo - R.README_BP_10000[0]) < 5e-8 and abs(hi - R.README_BP_10000[1]) < 5e-8
