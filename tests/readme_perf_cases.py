"""The reference README's PERFORMANCE-section problems (/root/reference/README.md:195-419) and the third-party solvers their
printed differences are taken against, restated for the tests (TEST INFRASTRUCTURE; data regenerated from the R snippets with
oracle/rrng.py, expected numbers copied from the README with their line numbers).

The README prints, for each problem, `range(<other package's coefficients> - <this package's>)`.  The other packages are absent
from this image (no R): what they compute is
  * glmnet (README.md:211-289): the elastic-net optimum on ITS lambda grid, to ITS convergence threshold.  Restated here from the
    published algorithm (Friedman, Hastie, Tibshirani 2010, J. Stat. Softw. 33(1), sections 2.1-2.5; glmnet's documented defaults
    nlambda = 100, lambda.min.ratio = 1e-4 (n >= p) / 0.01 (n < p), thresh = 1e-7, fdev = 1e-5, devmax = 0.999, mnlam = 5):
    `glmnet_grid_and_optimum` gives the grid and the EXACT optimum on it (scikit-learn's coordinate descent run to 1e-12);
    `glmnet_like_cd` runs naive cyclic coordinate descent with glmnet's own stopping rule max_j (delta beta_j)^2 < thresh on the
    standardised problem, i.e. an execution as inexact as glmnet's (not the same execution: glmnet's strong-rule screening and
    covariance updates are not restated);
  * quantreg::rq.fit (README.md:299-364): the optimum of the LAD linear programme -- `lad_lp` (SciPy HiGHS);
  * BP (README.md:370-419): the difference is taken against the TRUE coefficients: nothing third-party.
"""
import functools
import os

import numpy as np

from oracle.rrng import RRandom

HERE = os.path.dirname(os.path.abspath(__file__))

# README.md:238-242 (n = 10000, p = 1000) and :285-289 (n = 1000, p = 2000): rows glmnet-admm [lasso], glmnet-padmm[lasso], glmnet-admm [enet]
README_TALL = {"lasso": (-0.0002873333, 7.259293e-05), "padmm": (-0.0005554722, 7.382258e-05), "enet": (-0.0002195360, 8.176991e-05)}
README_WIDE = {"lasso": (-0.001518947, 0.002055109), "padmm": (-0.001898237, 0.002052009), "enet": (-0.001615556, 0.001948477)}
README_LAD_1000 = (-0.006989109, 0.006061505)          # README.md:331-333   rq.fit (method "br": the simplex, exact) - admm
README_LAD_5000 = (-0.003577610, 0.004135838)          # README.md:362-364   rq.fit(method = "fn": an interior-point approximation) - admm
README_BP_10000 = (-0.1575968, 0.3361001)              # README.md:417-419   beta_true - admm, n = 1000, p = 10000, nsig = 200
ENET_ALPHA = 0.6                                        # README.md:209,221


@functools.lru_cache(maxsize=None)
def lasso_data(n, p, m=100):
    """README.md:202-208 / :246-252: set.seed(123); b = c(runif(m), 0...); x = rnorm(n p, sd = 2); y = x b + rnorm(n)."""
    r = RRandom(123)
    b = np.concatenate([r.runif(m), np.zeros(p - m)])
    x = r.rnorm(n * p, sd=2.0).reshape((n, p), order="F")
    y = x @ b + r.rnorm(n)
    return x, y


@functools.lru_cache(maxsize=None)
def lad_data(n, p):
    """README.md:302-307 / :336-341: b = runif(p); x = rnorm(n p, sd = 2); y = x b + rnorm(n)."""
    r = RRandom(123)
    b = r.runif(p)
    x = r.rnorm(n * p, sd=2.0).reshape((n, p), order="F")
    y = x @ b + r.rnorm(n)
    return x, y


def _standardise(x, y):
    mx, sx = x.mean(0), x.std(0)                          # glmnet: 1/n variances
    my, sy = y.mean(), y.std()
    return (x - mx) / sx, (y - my) / sy, mx, sx, my, sy


def _path_ends(k, rsq, rsq0):
    """glmnet's early exit from the lambda loop (fdev, devmax; never before mnlam = 5 values)."""
    return k + 1 >= 5 and (rsq - rsq0 < 1e-5 * rsq or rsq > 0.999)


@functools.lru_cache(maxsize=None)
def glmnet_grid_and_optimum(n, p, alpha=1.0, nlambda=100):
    """(lambda grid, (p + 1) x nl coefficients on the original scale, row 0 = intercept): the optimum of
    (1 / 2n) ||y~ - X~ b||^2 + lambda~ (alpha ||b||_1 + (1 - alpha) / 2 ||b||^2) on glmnet's standardised problem (unit-variance
    y~: the ridge term is NOT scale-equivariant in y, and both packages scale y first -- Enet.cpp:99), mapped back."""
    from sklearn.linear_model import ElasticNet, Lasso
    x, y = lasso_data(n, p)
    xs, ys, mx, sx, my, sy = _standardise(x, y)
    ratio = 1e-4 if n >= p else 0.01
    lmax = np.abs(xs.T @ ys).max() / n / max(alpha, 1e-3)
    lam = lmax * ratio ** (np.arange(nlambda) / (nlambda - 1))
    gram = xs.T @ xs if n > p else False
    w = np.zeros(p)
    out, used, rsq0 = [], [], 0.0
    for k, l in enumerate(lam):
        kw = dict(alpha=l, fit_intercept=False, tol=1e-12, max_iter=500000, precompute=gram, warm_start=True)
        mdl = Lasso(**kw) if alpha == 1.0 else ElasticNet(l1_ratio=alpha, **kw)
        mdl.coef_ = w.copy()
        mdl.fit(xs, ys)
        w = mdl.coef_.copy()
        beta = w * sy / sx
        out.append(np.concatenate([[my - beta @ mx], beta]))
        used.append(l * sy)
        res = ys - xs @ w
        rsq = 1.0 - float(res @ res) / n
        if _path_ends(k, rsq, rsq0):
            break
        rsq0 = rsq
    return np.array(used), np.array(out).T


@functools.lru_cache(maxsize=None)
def glmnet_like_cd(n, p, alpha=1.0, nlambda=100, thresh=1e-7):
    """Naive cyclic coordinate descent (FHT 2010, section 2.1-2.2) with warm starts, active-set cycling (section 2.6) and
    glmnet's stopping rule: a sweep's largest (delta beta_j)^2 (standardised units, unit-variance y, so null deviance 1)
    below `thresh`.  Same outputs as glmnet_grid_and_optimum."""
    x, y = lasso_data(n, p)
    xs, r, mx, sx, my, sy = _standardise(x, y)
    xs = np.asfortranarray(xs)
    ratio = 1e-4 if n >= p else 0.01
    lmax = np.abs(xs.T @ r).max() / n / max(alpha, 1e-3)
    lam = lmax * ratio ** (np.arange(nlambda) / (nlambda - 1))
    w = np.zeros(p)
    ever = np.zeros(p, bool)
    out, used, rsq0 = [], [], 0.0
    for k, l in enumerate(lam):
        l1, l2 = l * alpha, l * (1.0 - alpha)

        def sweep(idx, r):
            dlx = 0.0
            for j in idx:
                xj = xs[:, j]
                g = xj @ r / n + w[j]
                u = abs(g) - l1
                nw = np.sign(g) * u / (1.0 + l2) if u > 0 else 0.0
                if nw != w[j]:
                    d = nw - w[j]
                    w[j] = nw
                    r = r - d * xj
                    dlx = max(dlx, d * d)
            return dlx, r

        while True:
            dlx, r = sweep(range(p), r)
            ever |= w != 0
            if dlx < thresh:
                break
            while True:
                dlx, r = sweep(np.nonzero(ever)[0], r)
                if dlx < thresh:
                    break
        beta = w * sy / sx
        out.append(np.concatenate([[my - beta @ mx], beta]))
        used.append(l * sy)
        rsq = 1.0 - float(r @ r) / n
        if _path_ends(k, rsq, rsq0):
            break
        rsq0 = rsq
    return np.array(used), np.array(out).T


def lad_lp(x, y):
    """The LAD linear programme min sum(u + v) s.t. X b + u - v = y, u, v >= 0 (what quantreg::rq.fit solves)."""
    import scipy.sparse as sp
    from scipy.optimize import linprog
    n, p = x.shape
    A = sp.hstack([sp.csr_matrix(x), sp.identity(n), -sp.identity(n)]).tocsc()
    c = np.concatenate([np.zeros(p), np.ones(2 * n)])
    res = linprog(c, A_eq=A, b_eq=y, bounds=[(None, None)] * p + [(0, None)] * (2 * n), method="highs")
    assert res.status == 0, res.message
    return res.x[:p], float(res.fun)


def lad_lp_n5000():
    """The LP optimum of the n = 5000, p = 1000 case: a fixture (9 minutes of HiGHS; tests/golden/make_readme_perf.py)."""
    f = np.load(os.path.join(HERE, "golden", "readme_lad_n5000_lp.npz"))
    return f["beta"], float(f["objective"])
