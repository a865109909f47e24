"""K-fold cross-validation entry point (admm_hip_lasso_cv; SURVEY.md section 8f row n4 -- not in the reference package).
The folds are ordinary fits, so they are held to the ordinary entry points bit for bit; the scoring kernel is held to
NumPy in double."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(n, p, m, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, p)) * 2
    b = np.zeros(p)
    b[:m] = rng.uniform(size=m)
    y = x @ b + rng.standard_normal(n)
    return np.asfortranarray(x), y


def _check(cv, x, y, fid, nfolds, make):
    n = x.shape[0]
    lam = cv.lambda_
    nl = lam.size
    # the full-data fit is the ordinary fit
    full = make(x, y).penalty(nlambda=nl).fit() if cv._auto else make(x, y).penalty(lambda_=lam).fit()
    assert np.array_equal(full.beta_dense, cv.fit.beta_dense) and np.array_equal(full.niter, cv.fit.niter)
    assert np.array_equal(full.lambda_, lam)
    for f in range(nfolds):
        tr, te = fid != f, fid == f
        direct = make(np.asfortranarray(x[tr]), y[tr]).penalty(lambda_=lam).fit()
        assert np.array_equal(direct.beta_dense, cv.fold_beta[f]), f"fold {f}: coefficients differ from a direct call on the training rows"
        assert np.array_equal(direct.niter, cv.fold_niter[f])
        b = cv.fold_beta[f].astype(np.float64)                       # (p + 1) x nlam, intercept first
        pred = b[0][None, :] + x[te] @ b[1:]
        mse = ((y[te][:, None] - pred) ** 2).mean(axis=0)
        assert np.allclose(cv.fold_mse[f], mse, rtol=1e-11, atol=0), (f, np.abs(cv.fold_mse[f] / mse - 1).max())
    cvm = cv.fold_mse.mean(axis=0)
    cvse = cv.fold_mse.std(axis=0, ddof=1) / np.sqrt(nfolds)
    assert np.allclose(cv.cvm, cvm, rtol=1e-13) and np.allclose(cv.cvse, cvse, rtol=1e-10)
    imin = int(np.argmin(cv.cvm))
    assert cv.idx_min == imin
    ok = np.nonzero(cv.cvm <= cv.cvm[imin] + cv.cvse[imin])[0]
    assert cv.idx_1se == int(ok[np.argmax(lam[ok])])
    assert lam[cv.idx_1se] >= lam[cv.idx_min]
    # the selected lambda is a sensible one: well inside the path for data with signal
    assert 0 < cv.idx_min


def test_cv_tall_lasso_default_folds():
    import admm_amd
    x, y = _data(600, 40, 8, 0)
    nfolds = 4
    cv = admm_amd.admm_lasso(x, y).penalty(nlambda=12).cv(nfolds=nfolds, keep_fold_beta=True)
    cv._auto = True
    fid = np.arange(600) % nfolds
    _check(cv, x, y, fid, nfolds, admm_amd.admm_lasso)


def test_cv_wide_enet_given_folds_and_grid():
    import admm_amd
    x, y = _data(90, 300, 6, 1)
    nfolds = 5
    rng = np.random.default_rng(5)
    fid = rng.permutation(np.arange(90) % nfolds).astype(np.int32)
    lam = np.geomspace(1.5, 0.05, 9)

    def make(xx, yy):
        return admm_amd.admm_enet(xx, yy)

    def make_a(xx, yy):
        m = admm_amd.admm_enet(xx, yy)
        orig = m.penalty
        m.penalty = lambda *a, **k: orig(*a, alpha=0.5, **k)
        return m

    cv = admm_amd.admm_enet(x, y).penalty(lambda_=lam, alpha=0.5).cv(nfolds=nfolds, fold_id=fid, keep_fold_beta=True)
    cv._auto = False
    _check(cv, x, y, fid, nfolds, make_a)


def test_cv_argument_checks():
    import admm_amd
    x, y = _data(50, 5, 2, 2)
    with pytest.raises(RuntimeError):
        admm_amd.admm_lasso(x, y).cv(nfolds=1)
    with pytest.raises(RuntimeError):
        admm_amd.admm_lasso(x, y).cv(nfolds=3, fold_id=np.full(50, 7, dtype=np.int32))
    with pytest.raises(RuntimeError):
        admm_amd.admm_lasso(x, y).cv(nfolds=3, fold_id=np.zeros(50, dtype=np.int32))     # folds 1, 2 empty


def test_multi_response_equals_separate_fits_tall_and_wide():
    """admm_hip_lasso_multi: every response is bit-identical to its own admm_lasso / admm_enet call (x standardised once,
    X'X formed once for the tall solver)."""
    import admm_amd
    rng = np.random.default_rng(3)
    for (n, p, kind) in ((500, 60, "lasso"), (70, 260, "enet")):
        x, y0 = _data(n, p, 6, 10 + n)
        Y = np.stack([y0, x @ rng.standard_normal(p) * 0.1 + rng.standard_normal(n), rng.standard_normal(n) * 3 + 1], axis=1)
        if kind == "lasso":
            mk = lambda yy: admm_amd.admm_lasso(x, yy).penalty(nlambda=9)
        else:
            mk = lambda yy: admm_amd.admm_enet(x, yy).penalty(nlambda=7, alpha=0.4)
        fits = mk(Y[:, 0]).fit_responses(Y)
        assert len(fits) == 3
        for j, f in enumerate(fits):
            one = mk(np.ascontiguousarray(Y[:, j])).fit()
            assert np.array_equal(one.lambda_, f.lambda_), (kind, j)
            assert np.array_equal(one.niter, f.niter), (kind, j, one.niter, f.niter)
            assert np.array_equal(one.beta_dense, f.beta_dense), (kind, j)


def test_cv_and_multi_edge_cases():
    """Folds whose training set flips the dispatch (full data tall, folds wide: each fold still equals the direct call on its
    rows), a single response through the multi-response entry, and leave-one-out on a tiny problem."""
    import admm_amd
    x, y = _data(62, 60, 4, 21)                       # n = p + 2: tall as a whole, wide once a fold is held out
    nfolds = 3
    cv = admm_amd.admm_lasso(x, y).penalty(nlambda=6).cv(nfolds=nfolds, keep_fold_beta=True)
    cv._auto = True
    _check(cv, x, y, np.arange(62) % nfolds, nfolds, admm_amd.admm_lasso)
    one = admm_amd.admm_lasso(x, y).penalty(nlambda=6).fit_responses(y.reshape(-1, 1))
    ref = admm_amd.admm_lasso(x, y).penalty(nlambda=6).fit()
    assert len(one) == 1 and np.array_equal(one[0].beta_dense, ref.beta_dense) and np.array_equal(one[0].niter, ref.niter)
    xs, ys = _data(12, 3, 2, 22)
    loo = admm_amd.admm_lasso(xs, ys).penalty(nlambda=4).cv(nfolds=12)
    assert loo.fold_mse.shape == (12, 4) and np.all(np.isfinite(loo.cvm)) and np.all(loo.cvse >= 0)
