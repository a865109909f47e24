"""GPU: admm_hip_dantzig -- the Dantzig selector of the reference's unbuilt src/TODO/ADMMDantzig.h / Dantzig.cpp restated on the
current ADMMBase::solve -- against oracle/solvers.py Dantzig decision by decision, and against
the linear programme it solves.  Double arithmetic throughout: the trace is compared directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(n, p, m, seed, scale=1.0, shift=0.0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, p)) * scale + shift
    b = np.zeros(p)
    b[rng.choice(p, m, replace=False)] = rng.standard_normal(m) * 2
    y = x @ b + 0.3 * rng.standard_normal(n) + 1.5
    return np.asfortranarray(x), y, b


def _compare(x, y, label, nlambda=8, ratio=0.05, standardize=True, intercept=True, maxit=3000, lam=None, rho=None):
    import admm_amd
    from oracle import entry
    m = admm_amd.admm_dantzig(x, y, intercept, standardize)
    m.penalty(lam, nlambda=nlambda, lambda_min_ratio=ratio) if lam is not None else m.penalty(nlambda=nlambda, lambda_min_ratio=ratio)
    m.opts(maxit=maxit, rho=rho)
    fit = m.fit(trace=True)
    d = {"trace": []}
    opts = dict(maxit=maxit, eps_abs=1e-5, eps_rel=1e-5, rho=-1.0 if rho is None else rho)
    ref = entry.admm_dantzig(x, y, lam, nlambda, ratio, standardize, intercept, opts, d)
    tr = np.asarray(d["trace"], dtype=np.float64)
    t = fit.trace
    assert t[0, 8] == -1
    t = t[1:]
    assert np.allclose(fit.lambda_, ref["lambda"], rtol=1e-12)
    assert abs(fit.stats["eig_est"] / d["solver"].lmax_est - 1) < 1e-9, (fit.stats["eig_est"], d["solver"].lmax_est)
    nrec = min(len(t), len(tr))
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    errs = dict(eps_p=rel(t[:nrec, 2], tr[:nrec, 2]), eps_d=rel(t[:nrec, 3], tr[:nrec, 3]), rp=rel(t[:nrec, 4], tr[:nrec, 4]),
                rd=rel(t[:nrec, 5], tr[:nrec, 5]), rho=rel(t[:nrec, 10], tr[:nrec, 10]))
    same = np.array_equal(t[:nrec, 0], tr[:nrec, 0]) and np.array_equal(t[:nrec, 1], tr[:nrec, 1]) and np.array_equal(t[:nrec, 8], tr[:nrec, 8])
    eb = np.abs(fit.beta_dense - ref["beta"]).max() / max(np.abs(ref["beta"]).max(), 1e-300)
    print(f"[dantzig {label}] niter {list(fit.niter)} (oracle {list(ref['niter'])}), {len(t)} decisions; trace vs oracle "
          f"{({k: float(f'{v:.1e}') for k, v in errs.items()})}, beta {eb:.1e}, x-update variant {fit.stats['xupdate_variant']}")
    assert list(fit.niter) == list(ref["niter"]), label
    assert len(t) == len(tr) and same, label
    assert max(errs.values()) < 1e-7 and eb < 1e-8, (label, errs, eb)
    return fit, ref


def test_tall_path_explicit_gram_four_flag_combinations():
    x, y, _ = _problem(600, 40, 6, 1, scale=2.0, shift=1.0)
    for std_, icpt in ((True, True), (False, True), (True, False), (False, False)):
        fit, ref = _compare(x, y, f"n=600 p=40 std={int(std_)} icpt={int(icpt)}", standardize=std_, intercept=icpt, maxit=3000 if icpt else 700)
        if icpt:
            assert max(fit.niter) <= 3000                      # converges on a comfortably tall, centred problem
        else:
            # uncentred columns with mean 1 / sd 2: X'X has one dominant eigenvalue, the loose Lanczos value undershoots it and the
            # linearised step does not majorise -- the restated algorithm runs into maxit, and the library does so on the same trajectory
            assert max(fit.niter) == 701 == max(ref["niter"])
        assert fit.stats["xupdate_variant"] == 0               # X'X formed explicitly (n > p, p <= 1000: ADMMDantzig.h:222)
        assert np.count_nonzero(fit.beta_dense[1:, 0]) == 0    # lambda_max: the null model (ADMMDantzig.h:134)


def test_operator_form_wide_and_large_p_with_maxit_exits():
    # p > n: every product is X'(X v); the restated algorithm does not converge here (tests/test_oracle_dantzig.py) -- the
    # library must fail to converge in exactly the same way: niter = maxit + 1, same trajectory
    x, y, _ = _problem(60, 90, 5, 2)
    fit, ref = _compare(x, y, "n=60 p=90 (operator form)", nlambda=4, ratio=0.2, maxit=400)
    assert fit.stats["xupdate_variant"] == 1
    assert 401 in list(fit.niter)
    # tall with p > 1000: operator form again
    x, y, _ = _problem(5200, 1030, 12, 3)
    fit, ref = _compare(x, y, "n=5200 p=1030 (operator form)", nlambda=3, ratio=0.3, maxit=600)
    assert fit.stats["xupdate_variant"] == 1


def test_user_lambda_given_rho_and_the_linear_programme():
    """The Dantzig selector is an LP: min 1'(u + w) s.t. -lambda <= X'(X (u - w) - y) <= lambda.  HiGHS on the standardised problem
    against the library's coefficients mapped back."""
    from scipy.optimize import linprog
    x, y, _ = _problem(500, 30, 5, 4)
    fit, ref = _compare(x, y, "n=500 p=30 user lambda, rho = 0.01", lam=[0.3, 0.1], standardize=False, intercept=False, rho=0.01, maxit=5000)
    assert max(fit.niter) <= 5000
    n, p = x.shape
    A = x.T @ x
    c = x.T @ y
    for l, lam in enumerate(fit.lambda_):
        il = lam * n                                            # internal lambda = lambda n / scaleY (scaleY = 1 without standardisation)
        G = np.vstack([np.hstack([A, -A]), np.hstack([-A, A])])
        h = np.concatenate([il + c, il - c])
        lp = linprog(np.ones(2 * p), A_ub=G, b_ub=h, bounds=[(0, None)] * (2 * p), method="highs")
        assert lp.status == 0
        b = fit.beta_dense[1:, l]
        viol = np.abs(A @ b - c).max() - il
        print(f"[dantzig vs LP] lambda {lam:g}: ||beta||_1 {np.abs(b).sum():.6f} (LP {lp.fun:.6f}), constraint violation {viol / il:.1e} of lambda")
        assert viol < 2e-3 * il and abs(np.abs(b).sum() / lp.fun - 1) < 3e-3


def test_arguments_and_printing():
    import admm_amd
    x, y, _ = _problem(50, 6, 2, 5)
    fit = admm_amd.admm_dantzig(x, y).penalty(nlambda=3).opts(maxit=50).fit()
    assert "ADMM Dantzig Selector fitting result" in repr(fit) and fit.beta_dense.dtype == np.float64
    with pytest.raises(ValueError):
        admm_amd.admm_dantzig(x, y).cv(3)
    with pytest.raises(RuntimeError):
        admm_amd.admm_dantzig(x[:, :2], y).penalty(nlambda=3).fit()       # fewer than 3 columns: no ncv = 3 Lanczos run
