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
    fit = admm_lasso(x, y).penalty(readme.LAMBDA).parallel().fit()       # default nthread = 2
    beta = fit.beta_dense[:, 0]
    assert relerr(beta, readme.LASSO_PARADMM) < TOL                      # README.md:66-88
    ref = entry.admm_parlasso(x, y, [readme.LAMBDA], 100, 1e-4, True, True, 2, entry.LASSO_OPTS)
    assert relerr(beta, ref["beta"][:, 0]) < TOL
    assert abs(int(fit.niter[0]) - int(ref["niter"][0])) <= max(3, 0.03 * ref["niter"][0])


@pytest.mark.parametrize("n,p,K", [(1500, 120, 3), (403, 300, 4), (900, 250, 2)])
def test_parlasso_path_vs_oracle(n, p, K):
    """Tall blocks (Cholesky branch), wide blocks (Woodbury branch, PADMMLasso.h:26-29), ragged last block."""
    from admm_amd import admm_lasso
    from oracle import entry
    x, y = synth_lasso(n, p, max(3, p // 10), seed=17)
    fit = admm_lasso(x, y).penalty(nlambda=6, lambda_min_ratio=0.01).parallel(K).opts(maxit=3000).fit()
    opts = dict(entry.LASSO_OPTS, maxit=3000)
    ref = entry.admm_parlasso(x, y, None, 6, 0.01, True, True, K, opts)
    assert np.allclose(fit.lambda_, ref["lambda"], rtol=1e-5)
    for j in range(6):
        assert relerr(fit.beta_dense[:, j], ref["beta"][:, j]) < 2 * TOL, j
    assert np.abs(fit.niter.astype(int) - ref["niter"].astype(int)).max() <= np.maximum(5, 0.05 * ref["niter"].max())
