import numpy as np

from oracle.tracecmp import compare_traces  # noqa: F401  (re-exported for the tools)


def relerr(a, b):
    """Norm-wise relative error per SURVEY.md section 8c: max|a-b| / max|b| (intercept included)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / den


def synth_lasso(n, p, m, seed=123, sd=2.0):
    """README.md:195-201 recipe scaled: X ~ N(0, sd^2), beta* = U(0,1) on the first m, y = X beta* + N(0,1)."""
    rng = np.random.default_rng(seed)
    b = np.concatenate([rng.uniform(size=m), np.zeros(p - m)])
    x = rng.standard_normal((n, p)) * sd
    y = x @ b + rng.standard_normal(n)
    return x, y


def exact_path_optimum(detail, lam_user, alpha=None):
    """Exact minimiser of the reference's objective on the oracle's standardised data (float64
    coordinate descent to 1e-13), mapped back with DataStd.recover: the yardstick for lambdas where
    the ADMM stopping rule is fragile.  Objective: 1/2||y - X b||^2 + lam_int * (alpha |b|_1 +
    (1-alpha)/2 |b|^2), lam_int = lam_user * n / scaleY (Lasso.cpp:99, ADMMEnet.h:24-40)."""
    from sklearn.linear_model import ElasticNet, Lasso
    solver, std = detail["solver"], detail["std"]
    X = np.asarray(solver.X, dtype=np.float64)
    Y = np.asarray(solver.Y, dtype=np.float64)
    n = X.shape[0]
    out = []
    for lu in np.atleast_1d(lam_user):
        li = float(np.float32(lu * n / np.float64(std.scaleY)))
        if alpha is None:
            m = Lasso(alpha=li / n, fit_intercept=False, tol=1e-13, max_iter=200000)
        else:
            m = ElasticNet(alpha=li / n, l1_ratio=float(alpha), fit_intercept=False, tol=1e-13, max_iter=200000)
        m.fit(X, Y)
        b0, coef = std.recover(m.coef_.astype(std.T))
        out.append(np.concatenate([[b0], coef]).astype(np.float64))
    return np.array(out).T


def assert_path_parity(beta_gpu, niter_gpu, ref, detail, tol=1e-4, alpha=None, n_tight_first=5):
    """Column-wise parity with the oracle.  A column must match to `tol` (norm-wise relative);
    the only accepted exception is a lambda where ADMM's loose stopping rule fired at a different
    iteration (counts differ by more than 2 there or at an earlier lambda of the warm-started
    path) -- there both are valid outputs of the reference's algorithm under rounding-level
    perturbation (the CPU oracle itself moves by this much when its Cholesky solve is replaced by
    an explicit inverse), so the GPU solution is then held to the exact optimum: it may not be
    further from it than twice the oracle's own worst column on the same path (the reference's
    accuracy: README.md:238-242,285-289 report 3e-4 .. 2e-3 against glmnet).  The first
    `n_tight_first` columns must be tight."""
    nl = beta_gpu.shape[1]
    ng = np.asarray(niter_gpu, dtype=int)
    nr = np.asarray(ref["niter"], dtype=int)
    loose = []
    for j in range(nl):
        e = relerr(beta_gpu[:, j], ref["beta"][:, j])
        if e < tol:
            continue
        assert j >= n_tight_first, (j, e)
        assert np.abs(ng[:j + 1] - nr[:j + 1]).max() > 2, (j, e, ng, nr)
        loose.append(j)
    if loose:
        exact = exact_path_optimum(detail, ref["lambda"], alpha)
        # the oracle's own worst distance to the optimum along this path = the solver's accuracy here
        worst_ref = max(relerr(ref["beta"][:, j], exact[:, j]) for j in range(1, nl))
        for j in loose:
            eg = relerr(beta_gpu[:, j], exact[:, j])
            assert eg <= max(2 * worst_ref, tol), (j, eg, worst_ref)
    return loose


# ---------------------------------------------------------------------------------------------------------------
# Tall path: parity judged on the decision trace, the oracle FOLLOWING the GPU through near-ties.
#
# ADMM's stopping rule (FADMMBase.h:213-217) and Goldstein's restart rule (:243) are threshold tests on float
# residuals.  At the default eps = 1e-5 those residuals are dominated by the rounding of the x-update: the NumPy
# oracle run with mathematically identical x-updates (oracle/variants.py: float LLT solve = the reference, float
# inverse, inverse rounded once from double, exact solve) changes the iteration count of ~40 of 185 lambdas between
# ANY two of them, the reference's float LLT solve against the exact solve included (tests/tools/flip_floor.py,
# tests/test_flip_floor.py), and on ill-conditioned problems one such flip can end a lambda thousands of iterations
# early.  Identical counts against one particular rounding are therefore not a property any implementation can have.
# What is required instead -- with no tolerance for "columns downstream of a count flip":
#   R1  the oracle is run in FOLLOW mode (oracle/solvers.py FADMM._decide): it takes the GPU's outcome of a threshold
#       test only where moving every entry of its OWN x and z by at most `band` (8) float ulps could have produced that
#       outcome (a near-tie rounding decides; the measured need is <= 4.1 ulps over all cases of tests/tools/flip_floor.py);
#       any other disagreement fails the test.  Every decision taken this way is counted and its distance reported.
#   R2  on that common trajectory the iteration counts are identical for every lambda and every beta column is within
#       `tol` (1e-4, the north_star bar) of the oracle's;
#   R3  the only columns that may exceed `tol` are those where the reference's own formula loses the digits: they must be
#       within `factor` (5) x the distance the oracle's two rounding variants, following the same decisions, have
#       drifted from it by that lambda
#       (e.g. maxit = 7 with rho five orders below the automatic value: z = (x + y/rho) - lambda/rho cancels 5 digits).
#       The number of such columns is returned; tests bound and print it.
def traced_fit(model, capacity=1 << 18):
    """Run a configured ADMM_Lasso / ADMM_Enet model through the prepared-problem entry points with the decision trace."""
    from admm_amd.api import LassoPlan
    plan = LassoPlan(model)
    plan.enable_trace(capacity)
    fit = plan.run()
    trace = plan.read_trace()
    plan.close()
    return fit, trace


def oracle_following(trace, x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha=None, mode="llt32", band=8.0):
    """The oracle (x-update rounding `mode`) following the decisions of `trace` (a libadmm_hip trace or another
    oracle's).  Returns (result dict, forced decisions, number of decisions consumed)."""
    from oracle import entry
    from oracle.variants import tall_variant
    t = np.asarray(trace, dtype=np.float64)
    if len(t) and t[0, 8] == -1:
        t = t[1:]                                       # libadmm_hip's cold-start record
    d = {"follow": t, "follow_band": band}
    with tall_variant(mode):
        if alpha is None:
            ref = entry.admm_lasso(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, d)
        else:
            ref = entry.admm_enet(x, y, lam, nlambda, lmin_ratio, standardize, intercept, alpha, opts, d)
    return ref, d["forced"], d["solver"].ndecisions


def col_err(a, b, floor):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-300)


def assert_tall_parity(beta, niter, trace, problem, tol=1e-4, factor=5.0, band=8.0, label=""):
    """problem: dict(x, y, lam, nlambda, lmin_ratio, standardize, intercept, opts, alpha) -- the oracle's arguments."""
    ref, forced, ndec = oracle_following(trace, band=band, **problem)          # R1 (raises FollowMismatch)
    t = np.asarray(trace)
    nrec = len(t) - (1 if len(t) and t[0, 8] == -1 else 0)
    assert ndec == nrec, (label, "the oracle consumed a different number of decisions than the GPU took", ndec, nrec)
    ng, nr = np.asarray(niter, dtype=int), np.asarray(ref["niter"], dtype=int)
    assert np.array_equal(ng, nr), (label, ng, nr)                               # R2
    nl = beta.shape[1]
    floor = 1e-3 * float(np.abs(ref["beta"]).max())          # null / tiny columns are measured on the scale of the path
    errs = [col_err(beta[:, j], ref["beta"][:, j], floor) for j in range(nl)]
    loose, yard = [], 0.0
    if max(errs) >= tol:                                                         # R3
        drift = np.zeros(nl)
        for mode in ("inv32", "exact"):
            v, _, _ = oracle_following(trace, band=1e9, mode=mode, **problem)
            drift = np.maximum(drift, [col_err(v["beta"][:, j], ref["beta"][:, j], floor) for j in range(nl)])
        drift = np.maximum.accumulate(drift)        # along a warm-started path the drift of a lambda carries into the next ones
        for j in range(nl):
            if errs[j] >= tol:
                yard = max(yard, float(drift[j]))
                assert errs[j] <= factor * drift[j], (label, f"lambda {j}: error {errs[j]:.2e} on a common trajectory; the oracle's own "
                                                             f"rounding variants differ by up to {drift[j]:.2e} by then")
                loose.append(j)
    fm = max([f["ulps"] for f in forced], default=0.0)
    first = forced[0]["lam"] if forced else None
    print(f"[parity {label}] {nrec} decisions, {len(forced)} near-ties taken from the GPU (largest needs {fm:.2f} ulps of rounding, "
          f"first at lambda {first}); niter identical; max beta err {max(errs):.2e}; columns beyond {tol:g}: {len(loose)} of {nl}"
          + (f" {loose} (oracle rounding variants differ by {yard:.2e})" if loose else ""))
    return dict(forced=forced, max_ulps=fm, first_forced_lambda=first, loose=loose, max_err=max(errs), errs=errs, ref=ref)
