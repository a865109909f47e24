"""GPU parity: row-block consensus Lasso ($parallel()) vs the oracle and the README paradmm column."""
import numpy as np
import pytest

from helpers import relerr, synth_lasso

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_readme_paradmm_column(readme_lasso_xy):
    from admm_amd import admm_lasso
    from oracle import entry, readme
    x, y = readme_lasso_xy
    from helpers import traced_parity
    prob = dict(x=x, y=y, lam=[readme.LAMBDA], nlambda=100, lmin_ratio=1e-4, standardize=True, intercept=True, opts=entry.LASSO_OPTS,
                alpha=None, nthread=2)
    # on the decision trace: iteration count IDENTICAL to the oracle's (339), beta within 1e-4 of it
    fit, rep = traced_parity(admm_lasso(x, y).penalty(readme.LAMBDA).parallel(), prob, TOL, label="README paradmm")     # default nthread = 2
    beta = fit.beta_dense[:, 0]
    assert relerr(beta, readme.LASSO_PARADMM) < TOL                      # README.md:66-88
    assert int(fit.niter[0]) == int(rep["ref"]["niter"][0]) == 339
    plain = admm_lasso(x, y).penalty(readme.LAMBDA).parallel().fit()
    assert np.array_equal(plain.beta_dense, fit.beta_dense) and list(plain.niter) == list(fit.niter)


@pytest.mark.parametrize("n,p,K", [(1500, 120, 3), (403, 300, 4), (900, 250, 2)])
def test_parlasso_path_vs_oracle(n, p, K):
    """Tall blocks (Cholesky branch), wide blocks (Woodbury branch, PADMMLasso.h:26-29), ragged last block -- judged on
    the decision trace (the oracle follows the GPU through rounding-level near-ties of the stopping test only)."""
    from admm_amd import admm_lasso
    from helpers import assert_followed_parity, traced_fit
    from oracle import entry
    x, y = synth_lasso(n, p, max(3, p // 10), seed=17)
    fit, trace = traced_fit(admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.01).parallel(K).opts(maxit=3000))
    opts = dict(entry.LASSO_OPTS, maxit=3000)
    prob = dict(x=x, y=y, lam=None, nlambda=6, lmin_ratio=0.01, standardize=True, intercept=True, opts=opts, alpha=None, nthread=K)
    rep = assert_followed_parity(fit.beta_dense, fit.niter, trace, prob, TOL, label=f"consensus n={n} p={p} K={K}")
    assert np.allclose(fit.lambda_, rep["ref"]["lambda"], rtol=1e-5)


def test_dist_entry_point_single_rank_matches_single_process():
    """The multi-process consensus path (RCCL communicator, global-moment standardisation, grouped
    all-reduce between `pack` and `z`) with ONE rank owning all row blocks must reproduce the
    single-process solver bit for bit.  (Real multi-rank runs need several GPUs; the protocol itself
    is covered by the world_size-2 gloo test in tests/test_dist_gloo.py.)"""
    from admm_amd import admm_lasso, dist
    x, y = synth_lasso(1201, 150, 12, seed=41)
    dist.init_comm(1, 0)
    try:
        for K in (1, 3):
            fit_d = dist.parlasso_dist(x, y, 1201, 150, K, nlambda=5, lambda_min_ratio=0.01, maxit=2000)
            fit_s = admm_lasso(x, y).penalty(nlambda=5, lambda_min_ratio=0.01).opts(maxit=2000)
            fit_s.nthread = K
            lib_fit = fit_s.fit() if K > 1 else None
            if lib_fit is not None:
                assert np.array_equal(fit_d.beta_dense, lib_fit.beta_dense)
                assert list(fit_d.niter) == list(lib_fit.niter)
            else:
                assert np.isfinite(fit_d.beta_dense).all() and fit_d.niter.min() >= 1
    finally:
        dist.finalize_comm()


def test_batched_worker_products_are_bit_identical_to_per_worker_launches(monkeypatch):
    """Several row blocks in one process: the workers' products of one kind go out as ONE launch (gemv_t_batch_kernel, the same
    workgroup body and partial order).  Both consensus branches, a remainder block: coefficients and counts bit for bit."""
    import admm_amd
    from helpers import synth_lasso
    for (n, p, K, nl, maxit) in ((900, 120, 4, 5, 300), (403, 300, 4, 3, 200), (250, 700, 3, 3, 150)):
        x, y = synth_lasso(n, p, 10, seed=n + p)
        out = []
        for flag in ("1", "0"):
            admm_amd.options.set(PAR_BATCH=flag)
            m = admm_amd.admm_lasso(x, y).penalty(nlambda=nl).opts(maxit=maxit)
            m.nthread = K
            out.append(m.fit())
        assert np.array_equal(out[0].beta_dense, out[1].beta_dense) and np.array_equal(out[0].niter, out[1].niter), (n, p, K)
