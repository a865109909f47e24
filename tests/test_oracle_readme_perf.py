"""CPU: the oracle against the numbers of the reference README's PERFORMANCE section (/root/reference/README.md:195-419) -- the
only numbers the reference holds for the WIDE solver (ADMMLassoWide / ADMMEnetWide), the Woodbury branch of the consensus
solver, the general (n > 2000) branch of ADMMLAD and BP at p = 10^4.  SURVEY.md section 8c lists these as "parity unpinned";
round 3's review asked for exactly this file.

Each README number is `range(<another package's coefficients> - <this package's>)` on data regenerated here from the R snippet
(oracle/rrng.py).  The other packages are restated in tests/readme_perf_cases.py (glmnet: grid + exact optimum by scikit-learn,
and an execution as INEXACT as glmnet's own -- naive coordinate descent with glmnet's stopping rule; rq.fit: the LAD linear
programme by HiGHS).  What can be pinned, and how tightly:

  * where THIS package's own error dominates the printed difference the README number is reproduced to 3-4 digits from the
    exact optimum: both `glmnet-padmm[lasso]` lower ends (tall: Cholesky blocks; wide: n = 1000, p = 2000 in two 500-row blocks
    = the WOODBURY branch, PADMMLasso.h:22-30), LAD against rq.fit's simplex (n = 1000: all 7 printed digits; n = 5000 = the
    general branch ADMMLAD.h:75-76 against the interior-point `fn`: 2 digits) and BP at p = 10^4 (all 7 digits);
  * where glmnet's own convergence error dominates (thresh = 1e-7 on (delta beta)^2 in standardised units is +-2e-3 in these
    coefficients for p > n) the README number cannot be reproduced without glmnet's exact execution; it is SANDWICHED: the
    oracle's distance from the exact optimum (+-2e-4 wide, +-3e-4 tall) is below the printed one, and the printed one is below
    what an execution with glmnet's stopping rule differs by.  For the wide serial solver this is a bound, not a digit-level
    pin, and DESIGN.md section 6 says so.
"""
import numpy as np
import pytest

import readme_perf_cases as R

pytestmark = pytest.mark.filterwarnings("ignore")


@pytest.fixture(scope="module")
def runs():
    """The oracle on the four Lasso-family README problems, on glmnet's grids (what `$penalty(lambdas1)` passes)."""
    from oracle import entry
    out = {}
    for shape, (n, p) in (("tall", (10000, 1000)), ("wide", (1000, 2000))):
        x, y = R.lasso_data(n, p)
        for kind, alpha in (("lasso", 1.0), ("enet", R.ENET_ALPHA)):
            lam, exact = R.glmnet_grid_and_optimum(n, p, alpha)
            lam2, cd = R.glmnet_like_cd(n, p, alpha)
            assert len(lam) == len(lam2) and np.allclose(lam, lam2)
            if kind == "lasso":
                ref = entry.admm_lasso(x, y, lam, 100, 1e-4, True, True, entry.LASSO_OPTS)
            else:
                ref = entry.admm_enet(x, y, lam, 100, 1e-4, True, True, alpha, entry.LASSO_OPTS)
            out[shape, kind] = dict(lam=lam, exact=exact, cd=cd, beta=ref["beta"].astype(np.float64), niter=ref["niter"])
    return out


def _range(a, b):
    d = a - b
    return float(d.min()), float(d.max())


@pytest.mark.parametrize("shape", ["tall", "wide"])
@pytest.mark.parametrize("kind", ["lasso", "enet"])
def test_serial_solver_difference_from_glmnet_is_explained(runs, shape, kind):
    """README.md:238-242 (tall) / :285-289 (wide), rows glmnet-admm [lasso] and glmnet-admm [enet]."""
    r = runs[shape, kind]
    readme = (R.README_TALL if shape == "tall" else R.README_WIDE)[kind]
    ex, cd = _range(r["exact"], r["beta"]), _range(r["cd"], r["beta"])
    print(f"[README {shape} {kind}] {len(r['lam'])} lambdas, {int(r['niter'].sum())} iterations: exact - admm {ex[0]:.3e} {ex[1]:.3e} | README (glmnet - admm) "
          f"{readme[0]:.3e} {readme[1]:.3e} | glmnet-like CD - admm {cd[0]:.3e} {cd[1]:.3e}")
    # the oracle is within the solver's tolerance of the exact optimum: a few 1e-4 of coefficients of size <= 1 (the README's own
    # printed distance from glmnet is the yardstick: never more than 1.6 x it, measured 0.08 .. 1.56)
    for e, rd in zip(ex, readme):
        assert abs(e) <= 1.6 * abs(rd) + 1e-6, (shape, kind, ex, readme)
    # ... and the printed distance is not more than an execution with glmnet's stopping rule differs by
    for c, rd in zip(cd, readme):
        assert abs(rd) <= 1.05 * abs(c), (shape, kind, cd, readme)
    if shape == "tall" and kind == "lasso":                  # glmnet's error is small on this end: 10 % of the printed number
        assert abs(ex[0] - readme[0]) <= 0.15 * abs(readme[0]), (ex, readme)


def test_tall_consensus_row_reproduces_the_readme(runs):
    """README.md:240 glmnet-padmm[lasso] min = -0.0005554722 (n = 10000, p = 1000, `$parallel()` = 2 row blocks, Cholesky
    branch): the consensus solver's own error dominates this end, so the exact optimum stands in for glmnet to 4 digits."""
    from oracle import entry
    x, y = R.lasso_data(10000, 1000)
    r = runs["tall", "lasso"]
    par = entry.admm_parlasso(x, y, r["lam"], 100, 1e-4, True, True, 2, entry.LASSO_OPTS)
    lo, hi = _range(r["exact"], par["beta"].astype(np.float64))
    print(f"[README tall padmm] exact - padmm {lo:.7e} {hi:.3e} | README {R.README_TALL['padmm']}")
    assert abs(lo - R.README_TALL["padmm"][0]) <= 2e-3 * abs(R.README_TALL["padmm"][0]), (lo, R.README_TALL["padmm"])
    assert 0 <= hi <= R.README_TALL["padmm"][1]


def test_wide_consensus_row_reproduces_the_readme_woodbury_branch(runs):
    """README.md:288 glmnet-padmm[lasso] min = -0.001898237 (n = 1000, p = 2000: two 500 x 2000 blocks -> the Woodbury branch of
    PADMMLasso_Worker::next_x, PADMMLasso.h:22-30, which no other reference number reaches): reproduced to 3 digits."""
    from oracle import entry
    x, y = R.lasso_data(1000, 2000)
    r = runs["wide", "lasso"]
    par = entry.admm_parlasso(x, y, r["lam"], 100, 1e-4, True, True, 2, entry.LASSO_OPTS)
    lo, hi = _range(r["exact"], par["beta"].astype(np.float64))
    print(f"[README wide padmm] {int(par['niter'].sum())} iterations: exact - padmm {lo:.7e} {hi:.3e} | README {R.README_WIDE['padmm']}")
    assert abs(lo - R.README_WIDE["padmm"][0]) <= 2e-3 * abs(R.README_WIDE["padmm"][0]), (lo, R.README_WIDE["padmm"])
    assert 0 <= hi <= R.README_WIDE["padmm"][1]


def test_lad_n1000_against_the_linear_programme():
    """README.md:331-333: range(rq.fit(x, y)$coefficients - admm_lad(x, y, intercept = FALSE)$fit()$beta[-1]), n = 1000, p = 500
    (hat-matrix branch).  `br` is the simplex: the LP's vertex.  All printed digits."""
    from oracle import entry
    x, y = R.lad_data(1000, 500)
    beta_lp, _ = R.lad_lp(x, y)
    ref = entry.admm_lad(x, y, False, entry.LAD_OPTS)
    lo, hi = _range(beta_lp, ref["beta"][1:])
    print(f"[README LAD n=1000] LP - admm {lo:.9f} {hi:.9f} | README {R.README_LAD_1000}; {ref['niter']} iterations")
    assert abs(lo - R.README_LAD_1000[0]) < 5e-9 and abs(hi - R.README_LAD_1000[1]) < 5e-9


def test_lad_n5000_general_branch_against_the_linear_programme():
    """README.md:362-364, n = 5000, p = 1000: n > 2000 takes X (X'X)^-1 X' (ADMMLAD.h:75-76) -- the branch no other reference
    number reaches.  The README's partner is rq.fit(method = "fn"), an interior-point approximation of the LP optimum: the
    exact LP optimum (fixture, tests/golden/make_readme_perf.py) reproduces the printed range to two digits."""
    from oracle import entry
    x, y = R.lad_data(5000, 1000)
    beta_lp, obj = R.lad_lp_n5000()
    assert abs(np.abs(y - x @ beta_lp).sum() - obj) < 1e-6 * obj          # the fixture belongs to these data
    ref = entry.admm_lad(x, y, False, entry.LAD_OPTS)
    lo, hi = _range(beta_lp, ref["beta"][1:])
    print(f"[README LAD n=5000] LP - admm {lo:.7f} {hi:.7f} | README {R.README_LAD_5000}; {ref['niter']} iterations")
    assert abs(lo - R.README_LAD_5000[0]) < 5e-5 and abs(hi - R.README_LAD_5000[1]) < 5e-5
    assert obj <= np.abs(y - x @ ref["beta"][1:]).sum() <= obj * (1 + 5e-3)   # ... and the ADMM iterate is optimal to the solver's tolerance (eps 1e-4: measured +3.3e-3)


def test_bp_p10000_against_the_truth():
    """README.md:417-419: range(beta_true - admm_bp(x, y)$fit()$beta), n = 1000, p = 10000, nsig = 200: all printed digits."""
    from oracle import entry, readme
    x, y, bt = readme.bp_data(1000, 10000, 200)
    ref = entry.admm_bp(x, y, entry.BP_OPTS)
    lo, hi = _range(bt, ref["beta"])
    print(f"[README BP p=10000] truth - admm {lo:.7f} {hi:.7f} | README {R.README_BP_10000}; {ref['niter']} iterations")
    assert abs(lo - R.README_BP_10000[0]) < 5e-8 and abs(hi - R.README_BP_10000[1]) < 5e-8
