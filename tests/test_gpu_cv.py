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


# ------------------------------------------------------------------------------------------------ folds as down-dates
def _std_train_f64(xt, yt, standardize, intercept):
    """DataStd (DataStd.h:89-155) on the training rows in float64: standardised x, y and the statistics."""
    n = xt.shape[0]
    flag = int(standardize) + 2 * int(intercept)
    mx, sx, my, sy = np.zeros(xt.shape[1]), np.ones(xt.shape[1]), 0.0, 1.0
    xs, ys = xt.copy(), yt.copy()
    if flag == 1:
        sy = np.sqrt(((yt - yt.mean()) ** 2).sum() / n); ys = yt / sy
        sx = np.sqrt(((xt - xt.mean(0)) ** 2).sum(0) / n); xs = xt / sx
    elif flag >= 2:
        my = yt.mean(); ys = yt - my; sy = np.sqrt((ys ** 2).sum() / n); ys = ys / sy
        mx = xt.mean(0); xs = xt - mx
        if flag == 3:
            sx = np.sqrt((xs ** 2).sum(0) / n); xs = xs / sx
    return xs, ys, mx, sx, my, sy


@pytest.mark.parametrize("standardize,intercept", [(True, True), (False, True), (True, False), (False, False)])
def test_cv_downdated_fold_system_against_float64(standardize, intercept):
    """The system a fold's tall solver gets when the folds are formed as down-dates (cv.hip: G_T = D^-1 (G_all - G_f - n_T delta
    delta') D^-1, X_T'y_T, the training rows' statistics) against DataStd + X'X + X'y of the training rows in float64, and
    against what a DIRECT fit forms (float standardisation, then the float matrix-core Gram of the training rows).
    This replaces bit-compatibility with a direct call for these fits -- the stated reason: the Gram is no longer one float
    accumulation over the training rows but the difference of two, corrected in double and rounded once.  The bar is the direct
    float Gram's OWN error against float64 (measured 0.5 - 1.2e-6 of the diagonal at n_T = 975: float products accumulated by
    the matrix cores): the down-dated one must stay within 2.5 x that (measured 1.5 - 1.9 x: two accumulations instead of one)
    and below 3e-6 outright; X'y, the means and the scales to float rounding."""
    import ctypes
    from admm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(17)
    n, p, nfolds, fold = 1300, 130, 4, 2
    x = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.5, 3.0, p) + rng.uniform(-2.0, 2.0, p))      # column means up to 2 sd
    y = x[:, :7] @ rng.uniform(size=7) + 0.5 * rng.standard_normal(n) + 3.0
    fid = rng.integers(0, nfolds, n).astype(np.int32)
    G = np.zeros((p, p), dtype=np.float32, order="F"); xy = np.zeros(p, dtype=np.float32)
    mx = np.zeros(p, dtype=np.float32); sx = np.zeros(p, dtype=np.float32); msy = np.zeros(2, dtype=np.float32)
    rc = lib.admm_hip_test_cv_fold_system(x.ctypes.data, y.ctypes.data, n, p, fid.ctypes.data, nfolds, fold, int(standardize), int(intercept),
                                          G.ctypes.data, xy.ctypes.data, mx.ctypes.data, sx.ctypes.data, msy.ctypes.data)
    assert rc == 0, lib.admm_hip_last_error()
    tr = fid != fold
    nt = int(tr.sum())
    xs, ys, mx64, sx64, my64, sy64 = _std_train_f64(x[tr], y[tr], standardize, intercept)
    G64, xy64 = xs.T @ xs, xs.T @ ys
    diag = np.abs(np.diag(G64)).max()
    eg = np.abs(G - G64).max() / diag
    exy = np.abs(xy - xy64).max() / (np.sqrt(nt) * np.linalg.norm(ys))
    # what a direct fit forms: the training rows standardised in float, Gram by the matrix-core kernel
    x32 = np.asfortranarray(xs.astype(np.float32))
    Gd = np.zeros((p, p), dtype=np.float32, order="F")
    assert lib.admm_hip_test_gram(x32.ctypes.data, nt, p, 1, 0, Gd.ctypes.data) == 0
    egd = np.abs(Gd - G64).max() / diag
    print(f"[cv down-date flags std={int(standardize)} icpt={int(intercept)}] Gram error {eg:.2e} of the diagonal (direct float Gram {egd:.2e}), X'y {exy:.2e}, "
          f"mean {np.abs(mx - mx64).max():.1e}, scale {np.abs(sx / sx64 - 1).max():.1e}")
    assert eg < 3e-6 and eg < 2.5 * egd and exy < 1e-6
    assert np.allclose(mx, mx64, rtol=0, atol=4e-7 * max(1.0, np.abs(mx64).max())) and np.allclose(sx, sx64, rtol=4e-7)
    assert abs(msy[0] - my64) <= 4e-7 * max(1.0, abs(my64)) and abs(msy[1] / sy64 - 1) <= 4e-7
    assert np.array_equal(G, G.T)                                           # symmetric by construction


def test_cv_with_downdated_folds_against_direct_calls(monkeypatch):
    """admm_hip_lasso_cv with the folds formed as down-dates (forced on: the automatic rule wants p >= 1024): the full-data fit is
    still the ordinary fit bit for bit; every fold is held to a direct call on its training rows -- not bit for bit any more
    (see the system test above) but as two runs of the same float algorithm on systems that differ by 1e-7: every coefficient
    column within 1e-4 of the direct call's on the problem's coefficient scale (measured 4e-6), the Lasso objective of the training
    rows within 1e-7 relative (measured 1.5e-9) (the objective is flat at the solution: it does not see which of the two stopped an iteration earlier), the
    held-out error table within 1e-4 (measured 2e-6)."""
    import admm_amd
    from helpers import col_err
    x, y = _data(1500, 60, 8, 21)
    x += np.linspace(-3, 3, 60)[None, :]                                     # column means that the folds' centring has to follow
    nfolds = 5
    fid = (np.arange(1500) * 7 % nfolds).astype(np.int32)
    admm_amd.options.set(CV_DOWNDATE="1")
    cv = admm_amd.admm_lasso(x, y).penalty(nlambda=15).cv(nfolds=nfolds, fold_id=fid, keep_fold_beta=True)
    admm_amd.options.set(CV_DOWNDATE="0")
    ref = admm_amd.admm_lasso(x, y).penalty(nlambda=15).cv(nfolds=nfolds, fold_id=fid, keep_fold_beta=True)
    full = admm_amd.admm_lasso(x, y).penalty(nlambda=15).fit()
    assert np.array_equal(full.beta_dense, cv.fit.beta_dense) and np.array_equal(full.niter, cv.fit.niter) and np.array_equal(full.lambda_, cv.lambda_)
    lam = cv.lambda_
    worst_b = worst_o = 0.0
    for f in range(nfolds):
        tr = fid != f
        xt, yt = x[tr], y[tr]
        direct = admm_amd.admm_lasso(np.asfortranarray(xt), yt).penalty(lambda_=lam).fit()
        assert np.array_equal(direct.beta_dense, ref.fold_beta[f])           # the direct mode is still the direct call
        bd, bc = direct.beta_dense.astype(np.float64), cv.fold_beta[f].astype(np.float64)
        floor = 1e-2 * np.abs(bd[1:]).max()
        worst_b = max(worst_b, max(col_err(bc[:, l], bd[:, l], floor) for l in range(lam.size)))
        # objective of the standardised training problem as the reference states it: 1/(2n) ||y - b0 - X b||^2 + lambda ||b||_1 on the
        # original scale is NOT what it minimises with standardize = TRUE; compare the fits through the loss both runs see
        for l in range(lam.size):
            sx = xt.std(0)
            od = 0.5 * ((yt - bd[0, l] - xt @ bd[1:, l]) ** 2).mean() + lam[l] * np.abs(bd[1:, l] * sx).sum()
            oc = 0.5 * ((yt - bc[0, l] - xt @ bc[1:, l]) ** 2).mean() + lam[l] * np.abs(bc[1:, l] * sx).sum()
            worst_o = max(worst_o, abs(oc - od) / od)
    dm = np.abs(cv.fold_mse / ref.fold_mse - 1).max()
    dn = np.abs(cv.fold_niter.astype(int) - ref.fold_niter.astype(int)).max()
    print(f"[cv down-dated folds vs direct calls] coefficient columns within {worst_b:.2e}, objective within {worst_o:.2e}, held-out MSE table within {dm:.2e}, "
          f"iteration counts differ by at most {dn}; idx_min {cv.idx_min} / {ref.idx_min}")
    assert worst_b < 1e-4 and worst_o < 1e-7 and dm < 1e-4
    assert cv.idx_min == ref.idx_min


def test_cv_downdate_is_automatic_where_the_gram_is_the_setup_cost(monkeypatch):
    """p >= 1024 and every training set taller than wide: the folds are formed as down-dates without being asked (stats of the
    full fit say so through t_gram covering the one-time base), and the table agrees with the direct mode to 1e-4."""
    import admm_amd
    x, y = _data(3300, 1030, 12, 33)
    admm_amd.options.set(CV_DOWNDATE=None)
    auto = admm_amd.admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.05).cv(nfolds=3)
    admm_amd.options.set(CV_DOWNDATE="1")
    forced = admm_amd.admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.05).cv(nfolds=3)
    admm_amd.options.set(CV_DOWNDATE="0")
    direct = admm_amd.admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.05).cv(nfolds=3)
    assert np.array_equal(auto.fold_mse, forced.fold_mse) and np.array_equal(auto.fold_niter, forced.fold_niter)      # automatic == forced on
    assert not np.array_equal(auto.fold_mse, direct.fold_mse)                                                           # and it is not the direct mode
    assert np.abs(auto.fold_mse / direct.fold_mse - 1).max() < 1e-4 and auto.idx_min == direct.idx_min
    assert np.array_equal(auto.fit.beta_dense, direct.fit.beta_dense)                                                   # the full-data fit is the ordinary fit either way
