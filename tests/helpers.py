import numpy as np


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
