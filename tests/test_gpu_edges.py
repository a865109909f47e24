"""GPU: edge cases through the C ABI vs the oracle -- minimal sizes, boundaries between solver branches,
single-lambda grids, the n == p switch (Lasso.cpp:73 sends n == p to the wide solver)."""
import numpy as np
import pytest

from helpers import relerr, synth_lasso

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_minimal_tall_needs_three_columns_for_the_lanczos_estimate():
    """p = 3 is the smallest Gram the reference's ncv = 3 Lanczos accepts; p = 2 needs a user rho."""
    from admm_amd import AdmmHipError, admm_lasso
    from oracle import entry
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal((12, 3)), rng.standard_normal(12)
    fit = admm_lasso(x, y).penalty(0.05).fit()
    ref = entry.admm_lasso(x, y, [0.05], 100, 1e-4, True, True, entry.LASSO_OPTS)
    assert relerr(fit.beta_dense[:, 0], ref["beta"][:, 0]) < TOL
    x2 = x[:, :2]
    with pytest.raises(AdmmHipError) as ei:
        admm_lasso(x2, y).penalty(0.05).fit()
    assert ei.value.code == 6                                    # ADMM_ERR_EIGS (the reference throws std::invalid_argument)
    fit = admm_lasso(x2, y).penalty(0.05).opts(rho=2.0).fit()
    ref = entry.admm_lasso(x2, y, [0.05], 100, 1e-4, True, True, dict(entry.LASSO_OPTS, rho=2.0))
    assert relerr(fit.beta_dense[:, 0], ref["beta"][:, 0]) < TOL


def test_square_problem_takes_the_wide_solver():
    from admm_amd import admm_lasso
    from oracle import entry
    x, y = synth_lasso(64, 64, 5, seed=9)
    from helpers import traced_parity
    prob = dict(x=x, y=y, lam=[0.3, 0.1], nlambda=100, lmin_ratio=1e-4, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=None)
    fit, _ = traced_parity(admm_lasso(x, y).penalty([0.3, 0.1]), prob, TOL, label="square problem")   # counts identical, columns 1e-4
    assert fit.stats["branch"] == 1


def test_single_auto_lambda_and_tiny_wide():
    from admm_amd import admm_enet, admm_lasso
    from oracle import entry
    x, y = synth_lasso(7, 40, 3, seed=13)
    fit = admm_lasso(x, y).penalty(nlambda=1).fit()              # one automatic lambda = lambda_max: null model
    assert fit.beta_dense.shape == (41, 1) and np.count_nonzero(fit.beta_dense[1:, 0]) == 0
    from helpers import traced_parity
    prob = dict(x=x, y=y, lam=[0.2], nlambda=100, lmin_ratio=0.01, standardize=True, intercept=True, opts=entry.LASSO_OPTS, alpha=0.3)
    traced_parity(admm_enet(x, y).penalty([0.2], alpha=0.3), prob, TOL, label="tiny wide enet")


def test_parallel_block_limits_and_single_block():
    from admm_amd import admm_lasso
    from oracle import entry
    x, y = synth_lasso(300, 26, 4, seed=17)
    from helpers import traced_parity
    K = 5                                                        # 5 < 26 / 5: the largest nthread $parallel() accepts here
    m = admm_lasso(x, y).penalty([0.2]).opts(maxit=3000)
    m.nthread = K
    prob = dict(x=x, y=y, lam=[0.2], nlambda=100, lmin_ratio=1e-4, standardize=True, intercept=True, opts=dict(entry.LASSO_OPTS, maxit=3000),
                alpha=None, nthread=K)
    traced_parity(m, prob, TOL, label="consensus K=5 of p=26")  # counts identical, column within 1e-4


def test_lad_and_bp_minimal_shapes():
    from admm_amd import admm_bp, admm_lad
    from oracle import entry
    rng = np.random.default_rng(23)
    x = rng.standard_normal((9, 8))
    y = rng.standard_normal(9)
    fit = admm_lad(x, y).fit()                                   # n = p + 1
    ref = entry.admm_lad(x, y, True, entry.LAD_OPTS)
    assert relerr(fit.beta, ref["beta"]) < 1e-3
    a = rng.standard_normal((1, 6))
    b = a @ np.array([0.0, 2.0, 0, 0, 0, 0])
    fit = admm_bp(a, b).fit()                                    # a single equation
    ref = entry.admm_bp(a, b, entry.BP_OPTS)
    assert relerr(np.asarray(fit.beta.todense()).ravel(), ref["beta"]) < 1e-3


def test_wide_gram_free_spectral_radius_matches_the_gram_based_value():
    """the wide solver's Lanczos products are X (X' v) on the stored X, no n x n Gram; same loose Ritz value as
    the Gram-based call (option WIDE_SPRAD=gram) to float rounding, hence the same rho and the same path."""
    import os
    from admm_amd import admm_lasso
    x, y = synth_lasso(300, 3000, 20, seed=51)
    free = admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.05).fit()
    from admm_amd import options
    with options(WIDE_SPRAD="gram"):
        gram = admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.05).fit()
    assert free.stats["branch"] == 1 and free.stats["t_gram"] == 0.0 and gram.stats["t_gram"] > 0.0
    assert abs(free.stats["eig_est"] - gram.stats["eig_est"]) < 2e-5 * gram.stats["eig_est"], (free.stats["eig_est"], gram.stats["eig_est"])
    for j in range(6):
        assert relerr(free.beta_dense[:, j], gram.beta_dense[:, j]) < 1e-4, j


def test_tall_memory_wall_is_reported_not_crashed_into():
    """The tall solver caches a p x p inverse: when that does not fit, ADMM_ERR_MEMORY with a pointer to $parallel(), instead
    of a failed allocation somewhere inside the setup (the test option TEST_FREE_BYTES pretends the device is nearly full)."""
    import os
    from admm_amd import AdmmHipError, admm_lasso
    x, y = synth_lasso(900, 300, 10, seed=52)
    from admm_amd import options
    with options(TEST_FREE_BYTES="200000"):
        with pytest.raises(AdmmHipError) as ei:
            admm_lasso(x, y).penalty(0.1).fit()
    assert ei.value.code == 9 and "parallel" in str(ei.value)
    fit = admm_lasso(x, y).penalty(0.1).fit()                      # and the same call works once there is room
    assert np.all(np.isfinite(fit.beta_dense))


def test_block_cache_is_bounded_reused_and_trimmed():
    """Round 5: released device blocks >= 32 MB are kept for the library's next call (hipMalloc / hipFree of GB-sized buffers cost up to
    0.25 s per call on some hosts) -- empty memory only, no solver state (`Lasso.cpp:74-76,126-129`).  After a fit some memory is still held;
    a second identical fit is bit-identical and holds no more than the first; admm_hip_trim_memory() gives everything back."""
    import torch
    from admm_amd import admm_lasso, load
    lib = load()

    def used():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return total - free

    rng = np.random.default_rng(5)
    n, p = 30000, 1500                                    # X: 180 MB as floats, 360 MB as the staged doubles
    x = rng.standard_normal((n, p)) * 2
    y = x[:, :10] @ rng.uniform(size=10) + rng.standard_normal(n)
    assert lib.admm_hip_trim_memory() == 0
    u0 = used()
    a = admm_lasso(x, y).penalty(nlambda=5).fit()
    u1 = used()
    b = admm_lasso(x, y).penalty(nlambda=5).fit()
    u2 = used()
    assert np.array_equal(a.beta_dense, b.beta_dense) and np.array_equal(a.niter, b.niter)
    assert u1 - u0 >= (32 << 20), "nothing was kept for re-use"
    assert u2 - u0 <= (u1 - u0) + (64 << 20), (u0, u1, u2)       # the second call re-used what the first left
    assert lib.admm_hip_trim_memory() == 0
    u3 = used()
    assert u3 - u0 <= (16 << 20), (u0, u3)
