"""GPU parity: wide (n <= p) Lasso / Elastic-net (linearised ADMM with active-set iterations) vs the oracle.

The reference holds no known-answer vector for this solver (README.md:285-289 only gives ranges
against glmnet), so the oracle -- pinned on the five README vectors with which it shares all code
-- is the reference here ("parity unpinned" in oracle/__init__.py)."""
import numpy as np
import pytest

from helpers import assert_followed_parity, relerr, synth_lasso, traced_fit

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _problem(x, y, nl, lmr, alpha=None):
    from oracle import entry
    return dict(x=x, y=y, lam=None, nlambda=nl, lmin_ratio=lmr, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=alpha)


@pytest.mark.parametrize("n,p,m,screen", [(300, 2000, 20, ""), (257, 1031, 10, ""), (500, 500, 15, ""), (300, 2000, 20, "16"), (1500, 4000, 25, "16"), (1500, 4000, 25, "8")])
def test_wide_lasso_path_vs_oracle(n, p, m, screen):
    """12-lambda path judged on the decision trace: the oracle follows the GPU through rounding-level near-ties of the
    stopping test and of the rho adaptation only (helpers.assert_followed_parity); counts identical, every column 1e-4.
    screen = "1": with the regular steps screened (forced on at this size): follow + stepwise hold it like the default."""
    from admm_amd import admm_lasso, options
    from helpers import traced_parity
    x, y = synth_lasso(n, p, m, seed=29)
    lmr = 0.01 if n < p else 1e-4                                # R default (R/30_admm_lasso.R:43)
    if screen:
        options.set(WIDE_SCREEN=screen)                          # (reset after every test: conftest.py)
    # the follow rule AND the stepwise rule on the iterate dump (helpers.traced_parity -> wide_stepwise)
    fit, rep = traced_parity(admm_lasso(x, y).penalty(nlambda=12, lambda_min_ratio=lmr), _problem(x, y, 12, lmr), TOL, label=f"wide n={n} p={p}")
    assert fit.stats["branch"] == 1 and fit.stats["xupdate_variant"] == {"": 0, "16": 1, "8": 2}[screen]
    ref = rep["ref"]
    assert np.allclose(fit.lambda_, ref["lambda"], rtol=1e-5)
    # support agreement
    for j in range(12):
        assert np.array_equal(np.abs(fit.beta_dense[1:, j]) > 1e-5, np.abs(ref["beta"][1:, j]) > 1e-5), j
    fit2 = admm_lasso(x, y).penalty(nlambda=12, lambda_min_ratio=lmr).fit()
    assert np.array_equal(fit2.beta_dense, fit.beta_dense) and list(fit2.niter) == list(fit.niter)


def test_wide_spectral_radius_estimate():
    from admm_amd import admm_lasso
    from oracle import entry
    x, y = synth_lasso(300, 2000, 20, seed=29)
    fit = admm_lasso(x, y).penalty(nlambda=2, lambda_min_ratio=0.5).fit()
    d = {}
    entry.admm_lasso(x, y, None, 2, 0.5, True, True, entry.LASSO_OPTS, d)
    assert abs(fit.stats["eig_est"] - float(d["solver"].sprad)) < 1e-4 * float(d["solver"].sprad)


@pytest.mark.parametrize("screen", ["", "16", "8"])
def test_wide_enet_path_vs_oracle(screen):
    from admm_amd import admm_enet, options
    x, y = synth_lasso(250, 900, 12, seed=31)
    from helpers import traced_parity
    if screen:
        options.set(WIDE_SCREEN=screen)
    traced_parity(admm_enet(x, y).penalty(nlambda=10, lambda_min_ratio=0.01, alpha=0.5), _problem(x, y, 10, 0.01, alpha=0.5), TOL, label="wide enet")


def test_wide_user_lambda_and_maxit():
    from admm_amd import admm_lasso
    from oracle import entry
    x, y = synth_lasso(120, 400, 8, seed=37)
    lam = [0.8, 0.2, 0.05]
    from helpers import traced_parity
    for maxit in (7, 10000):
        prob = dict(x=x, y=y, lam=lam, nlambda=100, lmin_ratio=0.01, standardize=True, intercept=True, opts=dict(entry.LASSO_OPTS, maxit=maxit), alpha=None)
        fit, _ = traced_parity(admm_lasso(x, y).penalty(lam).opts(maxit=maxit), prob, TOL, label=f"wide user lambda maxit {maxit}")
        if maxit < 100:
            assert list(fit.niter) == [8, 8, 8]


def _fit_env(x, y, nl, maxit, **env):
    """Fit with kernel-variant options of the calling thread (admm_amd.options; read at plan creation)."""
    from admm_amd import admm_lasso, options
    with options(**env):
        return admm_lasso(x, y).penalty(nlambda=nl).opts(maxit=maxit).fit()


def _traced_env(x, y, nl, maxit, label, **env):
    """Traced fit under kernel-variant knobs, judged by the trace rule against the oracle (counts identical, columns 1e-4)."""
    from admm_amd import admm_lasso, options
    from helpers import traced_parity
    from oracle import entry
    with options(**env):
        prob = dict(x=x, y=y, lam=None, nlambda=nl, lmin_ratio=0.01, standardize=True, intercept=True, opts=dict(entry.LASSO_OPTS, maxit=maxit), alpha=None)
        return traced_parity(admm_lasso(x, y).penalty(nlambda=nl, lambda_min_ratio=0.01).opts(maxit=maxit), prob, TOL, label=label)[0]


def test_wide_large_n_lds_modes():
    """The x-update stages 2 n floats of t in LDS.  n = 9000 needs 72 KB (> the 64 KB default: opt-in attribute, up to
    160 KB on gfx950) and n = 21000 needs 168 KB (> the opt-in limit: t goes through global memory instead).  Round 1
    launched both sizes with an unchecked LDS request and failed every launch.  The global-t mode is bit-identical to
    the LDS mode where both exist, and the large sizes reproduce the oracle."""
    from oracle import entry
    x, y = synth_lasso(3000, 3600, 15, seed=41)
    a = _fit_env(x, y, 4, 40, WIDE_FUSE="0")
    b = _fit_env(x, y, 4, 40, WIDE_TGLOBAL="1")
    assert np.array_equal(a.beta_dense, b.beta_dense) and list(a.niter) == list(b.niter)
    x, y = synth_lasso(9000, 9500, 20, seed=43)
    a = _traced_env(x, y, 3, 25, "wide n=9000 (LDS opt-in)")                      # trace rule: counts identical to the oracle's, columns 1e-4
    b = _fit_env(x, y, 3, 25, WIDE_TGLOBAL="1")
    assert np.array_equal(a.beta_dense, b.beta_dense) and list(a.niter) == list(b.niter)
    x, y = synth_lasso(21000, 21500, 20, seed=47)
    _traced_env(x, y, 2, 12, "wide n=21000 (t through global memory)")


@pytest.mark.parametrize("n,p", [(5000, 5600), (7600, 8000)])
def test_wide_fused_x_update_up_to_8192_rows(n, p):
    """n in (4096, 8192]: the column just dotted with t still fits the registers of one wave (24 / 32 float4 per lane), so the
    x-update also gathers A x (two launches per iteration instead of three).  Same algorithm: iteration counts and
    coefficients of the three-launch path (different order of the partial sums: equal up to rounding) and of the oracle."""
    from oracle import entry
    x, y = synth_lasso(n, p, 20, seed=n)
    a = _traced_env(x, y, 3, 30, f"wide fused n={n}")                              # each variant is its own execution: both held to the oracle
    b = _traced_env(x, y, 3, 30, f"wide three launches n={n}", WIDE_FUSE="0")
    assert a.stats["xupdate_launches"] > 0
    for j in range(3):
        assert relerr(a.beta_dense[:, j], b.beta_dense[:, j]) < 1e-4, j


@pytest.mark.parametrize("n,p,cgroups", [(600, 9000, ""), (1500, 4000, ""), (3000, 5000, ""), (257, 1031, ""), (600, 9000, "1"), (5000, 5600, ""), (1500, 4000, "2")])
def test_persistent_active_set_stretch(n, p, cgroups):
    """wide_rows_persist_kernel (round 4): the stretch on a 2-D grid of workgroups -- R row groups of 256 rows x C column groups
    (R C <= 32): partials of A x exchanged inside a row group, partial dots of the active columns inside a column group together
    with the row groups' norm shares; the next t formed speculatively (same rho) and formed again after a rho change; the column
    slices resident in registers.  Its own execution (other order of the sums inside the two mat-vecs): held to the oracle by the
    trace rule AND by the stepwise rule on its iterate dump (helpers.traced_parity), and within summation rounding of the two-launch
    path.  n = 600: 3 x 10 workgroups; n = 1500: 6 x 5; n = 3000 / 5000: 12 x 2 / 20 x 1 (beyond the 2048 rows the round-3 stretch
    stops at); n = 257: a second row group that owns ONE row, 2 x 16; C forced to 1 / 2: no / fewer column groups."""
    x, y = synth_lasso(n, p, 15, seed=n + p)
    env = {}                                                 # the stretch is the default (WIDE_PERSIST=0 switches it off)
    if cgroups:
        env["WIDE_ROWS_C"] = cgroups
    a = _traced_env(x, y, 10, 10000, f"wide 2-D stretch n={n} C={cgroups or 'auto'}", **env)
    b = _fit_env(x, y, 10, 10000, WIDE_PERSIST="0")
    assert int(a.stats["persist_iter"]) > 0.5 * int(a.stats["total_iter"]), (a.stats["persist_iter"], a.stats["total_iter"])
    for j in range(10):
        assert relerr(a.beta_dense[:, j], b.beta_dense[:, j]) < 1e-4, j


def _wide_fit(x, y, enet, **opt):
    from admm_amd import admm_enet, admm_lasso, options
    with options(**opt):
        m = admm_enet(x, y) if enet else admm_lasso(x, y)
        m = m.penalty(nlambda=10, lambda_min_ratio=0.02, alpha=0.6) if enet else m.penalty(nlambda=10, lambda_min_ratio=0.02)
        return traced_fit(m)


@pytest.mark.parametrize("fmt", ["16", "8"])
@pytest.mark.parametrize("n,p,enet", [(300, 3000, False), (900, 5000, True), (1500, 6000, False), (2000, 9000, True), (3000, 5000, False), (5000, 5600, False), (7600, 8000, True)])
def test_screened_regular_steps_are_bit_identical(n, p, enet, fmt):
    """The regular steps' safe screen (wide_x_kernel: a product with the fp16-rounded column proves "stays zero" for all but a few
    columns -- or, fmt 8, with its linear 8-bit code; every other column takes the exact float step) must not change a single bit: the same path with WIDE_SCREEN=16 / 8 and =0
    -- coefficients, iteration counts and every record of the decision trace (residuals, thresholds, rho) identical.  All five
    register layouts of the fused x-update (n <= 1024 / 2048 / 4096 / 6144 / 8192), lasso (double compare) and elastic net (float
    compare).  The screened run is then also held to the oracle by the trace rule like every other variant."""
    x, y = synth_lasso(n, p, 15, seed=211 + n)
    on, tr_on = _wide_fit(x, y, enet, WIDE_SCREEN=fmt)
    off, tr_off = _wide_fit(x, y, enet, WIDE_SCREEN="0")
    assert on.stats["xupdate_variant"] == (2 if fmt == "8" else 1) and off.stats["xupdate_variant"] == 0 and on.stats["branch"] == 1
    assert list(on.niter) == list(off.niter)
    assert np.array_equal(on.beta_dense, off.beta_dense)
    assert tr_on.shape == tr_off.shape and np.array_equal(tr_on, tr_off)
    assert sum(on.niter) > 100


@pytest.mark.parametrize("fmt", ["16", "8"])
@pytest.mark.parametrize("case", ["huge", "tiny", "mixed", "nan_free_inf_round"])
def test_screen_with_values_fp16_cannot_hold(case, fmt):
    """Columns whose entries overflow fp16 (|x| > 65504: stored as zero in the copy, the whole entry counted as rounding error), fall
    into its subnormal range or underflow to zero, without standardisation: still bit-identical to the unscreened run."""
    rng = np.random.default_rng(5)
    n, p = 400, 3000
    x = rng.standard_normal((n, p))
    scale = np.ones(p)
    if case == "huge":
        scale[::3] = 3e5
    elif case == "tiny":
        scale[::2] = 1e-6
        scale[1::4] = 3e-9
    elif case == "mixed":
        scale = 10.0 ** rng.uniform(-8, 6, p)
    else:
        x[rng.random((n, p)) < 0.01] *= 7e4                       # isolated entries beyond 65504 inside ordinary columns
    x = x * scale
    beta = np.zeros(p)
    beta[rng.choice(p, 12, replace=False)] = rng.standard_normal(12) / scale[:12].mean()
    y = x @ beta + 0.1 * rng.standard_normal(n) * np.abs(x @ beta).mean()
    from admm_amd import admm_lasso, options
    fits = {}
    for scr in (fmt, "0"):
        with options(WIDE_SCREEN=scr):
            fits[scr] = traced_fit(admm_lasso(x, y, standardize=False).penalty(nlambda=8, lambda_min_ratio=0.05).opts(maxit=400))
    (on, tr_on), (off, tr_off) = fits[fmt], fits["0"]
    assert on.stats["xupdate_variant"] == (2 if fmt == "8" else 1)
    assert list(on.niter) == list(off.niter) and np.array_equal(on.beta_dense, off.beta_dense) and np.array_equal(tr_on, tr_off)


def test_screen_picks_the_copy_by_the_columns_at_hand():
    """The default's choice (WIDE_SCREEN=auto: the same rule at any size): Gaussian-looking columns take the 8-bit code, columns with
    outliers -- which waste a linear code's range: its bounds would send a large share of the columns down the exact path -- the fp16
    copy.  Either way the result is the unscreened one."""
    from admm_amd import admm_lasso, options
    rng = np.random.default_rng(3)
    n, p = 600, 4000
    x, y = synth_lasso(n, p, 15, seed=77)
    xo = x.copy()
    xo[rng.integers(0, n, p), np.arange(p)] *= 400.0              # one outlier per column
    for data, want in ((x, 2), (xo, 1)):
        with options(WIDE_SCREEN="auto"):
            a = admm_lasso(data, y).penalty(nlambda=6, lambda_min_ratio=0.05).fit()
        with options(WIDE_SCREEN="0"):
            b = admm_lasso(data, y).penalty(nlambda=6, lambda_min_ratio=0.05).fit()
        assert a.stats["xupdate_variant"] == want and b.stats["xupdate_variant"] == 0
        assert np.array_equal(a.beta_dense, b.beta_dense) and list(a.niter) == list(b.niter)
