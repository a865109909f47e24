"""GPU: the hand-written MFMA Gram kernel vs the rocBLAS path, end to end through the tall solver."""
import os

import numpy as np
import pytest

from helpers import relerr, synth_lasso

pytestmark = pytest.mark.gpu


def test_mfma_gram_matches_library_gram_end_to_end():
    """p = 2100 -> 17 x 17 tiles / 2 = 153 tiles >= 128: the MFMA path is taken by default (ragged last tile)."""
    from admm_amd import admm_lasso
    x, y = synth_lasso(6000, 2100, 50, seed=61)
    lam = [0.3, 0.05]
    os.environ["ADMM_HIP_GRAM"] = "rocblas"
    try:
        ref = admm_lasso(x, y).penalty(lam).fit()
    finally:
        del os.environ["ADMM_HIP_GRAM"]
    fit = admm_lasso(x, y).penalty(lam).fit()
    # same Lanczos estimate (sets rho) to float rounding, same path
    assert abs(fit.stats["eig_est"] - ref.stats["eig_est"]) < 1e-5 * ref.stats["eig_est"]
    assert np.abs(fit.niter.astype(int) - ref.niter.astype(int)).max() <= 2
    for j in range(2):
        assert relerr(fit.beta_dense[:, j], ref.beta_dense[:, j]) < 1e-4
