"""The compiled C restatements of the wide / consensus / LAD / BP loops (oracle/c/admm_loops_cpu.c, bench.py's cpu_baseline legs)
against the NumPy oracle (oracle/solvers.py, pinned on the README vectors): same decisions, same coefficients -- on fixed-maxit
runs (no stopping decision to flip) to rounding, on converged runs within what one flipped near-tie may cost."""
import numpy as np
import pytest

from helpers import relerr, synth_lasso


def _same_decisions(tr_c, tr_np, rho_col_c=10, rho_col_np=10):
    a, b = np.asarray(tr_c), np.asarray(tr_np, dtype=np.float64)
    assert len(a) == len(b), (len(a), len(b))
    assert np.array_equal(a[:, 0], b[:, 0]) and np.array_equal(a[:, 1], b[:, 1])
    assert np.array_equal(a[:, 8].astype(int), b[:, 8].astype(int))
    assert np.allclose(a[:, rho_col_c], b[:, rho_col_np], rtol=1e-12)
    return a, b


def _same_scalars(a, b, tight=1e-4):
    """thresholds to `tight`; r_p to 1 %; r_d to 10 % while it is above its threshold (below, rho ||z - z_old|| is single-ulp
    flips of a few z entries -- tests/test_oracle_c.py)"""
    assert np.allclose(a[:, 2], b[:, 2], rtol=tight) and np.allclose(a[:, 3], b[:, 3], rtol=tight)
    assert np.allclose(a[:, 4], b[:, 4], rtol=1e-2)
    big = b[:, 5] > b[:, 3]
    assert np.allclose(a[big, 5], b[big, 5], rtol=0.1)


@pytest.mark.parametrize("nthreads", [1, 3])
def test_wide_loop_c_vs_numpy(nthreads):
    from oracle import cloops, entry
    x, y = synth_lasso(150, 900, 12, seed=5)
    opts = dict(entry.LASSO_OPTS, maxit=45)                    # regular steps 0 / 3 / 15, active-set steps between them, rho adaptation, all exits at maxit
    trn, trc = {"trace": []}, []
    ref = entry.admm_lasso(x, y, None, 5, 0.05, True, True, opts, trn)
    got = cloops.admm_lasso_wide_c(x, y, None, 5, 0.05, True, True, opts, nthreads=nthreads, trace=trc)
    assert np.allclose(got["lambda"], ref["lambda"])
    a, b = _same_decisions(trc, trn["trace"])
    _same_scalars(a, b)
    assert list(got["niter"]) == list(ref["niter"])
    assert relerr(got["beta"], ref["beta"]) < 2e-5
    assert got["nnz_sum"] == int(np.sum(trn["solver"].trace_nnz))
    # to convergence: the float dot products add up in another order than OpenBLAS's, a near-tie may flip a count late in the path
    opts = dict(entry.LASSO_OPTS)
    ref = entry.admm_lasso(x, y, None, 6, 0.1, True, True, opts)
    got = cloops.admm_lasso_wide_c(x, y, None, 6, 0.1, True, True, opts, nthreads=nthreads)
    assert np.abs(got["niter"][:3].astype(int) - ref["niter"][:3].astype(int)).max() <= 2, (got["niter"], ref["niter"])
    for j in range(6):
        assert relerr(got["beta"][:, j], ref["beta"][:, j]) < 2e-3, j


@pytest.mark.parametrize("shape,K,nthreads", [((900, 60), 3, 1), ((320, 1500), 4, 2), ((320, 1500), 4, 9)])
def test_consensus_loop_c_vs_numpy(shape, K, nthreads):
    """Cholesky branch (tall blocks) and Woodbury branch (wide blocks, PADMMLasso.h:23-30); 9 threads over 4 workers also puts
    threads inside the workers' products."""
    from oracle import cloops, entry
    x, y = synth_lasso(shape[0], shape[1], 10, seed=8)
    opts = dict(entry.LASSO_OPTS, maxit=60)
    trn, trc = {"trace": []}, []
    ref = entry.admm_parlasso(x, y, None, 3, 0.3, True, True, K, opts, trn)
    got = cloops.admm_parlasso_c(x, y, None, 3, 0.3, True, True, K, opts, nthreads=nthreads, trace=trc)
    a, b = _same_decisions(trc, trn["trace"], 6, 6)
    _same_scalars(a, b)
    assert list(got["niter"]) == list(ref["niter"])
    assert abs(got["rho"] - trn["solver"].rho) < 1e-12 * got["rho"]
    assert relerr(got["beta"], ref["beta"]) < 5e-5


def test_lad_and_bp_loops_c_vs_numpy():
    from oracle import cloops, entry
    rng = np.random.default_rng(3)
    n, p = 2400, 40                                            # n > 2000: the oracle takes the general branch too
    x = rng.standard_normal((n, p)) * 2 + 0.3
    y = x @ rng.uniform(size=p) + rng.standard_t(3, size=n) + 1.5
    for nt in (1, 3):
        trn, trc = {"trace": []}, []
        ref = entry.admm_lad(x, y, True, entry.LAD_OPTS, trn)
        got = cloops.admm_lad_c(x, y, True, entry.LAD_OPTS, nthreads=nt, trace=trc)
        assert got["niter"] == ref["niter"], (got["niter"], ref["niter"])
        a, b = _same_decisions(trc, trn["trace"])
        assert np.allclose(a[:, 2:7], b[:, 2:7], rtol=1e-6)
        assert relerr(got["beta"], ref["beta"]) < 1e-9
    A = rng.standard_normal((120, 400))
    bt = np.zeros(400); bt[rng.choice(400, 12, replace=False)] = rng.uniform(size=12)
    for nt in (1, 3):
        trn, trc = {"trace": []}, []
        ref = entry.admm_bp(A, A @ bt, entry.BP_OPTS, trn)
        got = cloops.admm_bp_c(A, A @ bt, entry.BP_OPTS, nthreads=nt, trace=trc)
        assert got["niter"] == ref["niter"], (got["niter"], ref["niter"])
        a, b = _same_decisions(trc, trn["trace"])
        assert relerr(got["beta"], ref["beta"]) < 1e-9
        assert abs(got["rho"] - trn["solver"].rho) < 1e-12 * got["rho"]


def test_readme_lad_and_bp_through_the_c_loops(readme_lasso_xy):
    """README known answers through the compiled loops (LAD: general branch instead of the hat matrix the reference caches at
    n = 100 -- the same projection; README.md:139-161, 165-182)."""
    from oracle import cloops, entry, readme
    x, y = readme_lasso_xy
    r = cloops.admm_lad_c(x, y, False, entry.LAD_OPTS)
    assert relerr(r["beta"][1:], readme.LAD_ADMM) < 1e-4
    xb, yb, bt = readme.bp_data()
    r = cloops.admm_bp_c(xb, yb, entry.BP_OPTS)
    e = bt - r["beta"]
    assert abs(e.min() - readme.BP_RANGE[0]) < 1e-6 and abs(e.max() - readme.BP_RANGE[1]) < 1e-6


def test_sharing_bp_loop_c_vs_numpy():
    """The compiled restatement of the column-block sharing basis pursuit (bench.py's cpu_baseline of the parbp config) against the
    NumPy class it follows (oracle/solvers.py SharingBP): the README basis-pursuit data (README.md:217-246) and a ragged random
    problem, 2 .. 7 blocks, one and three threads -- same iteration counts, the same regular / active-set schedule and decisions,
    every recorded threshold and residual to 1e-9 of its column's largest value, coefficients to 1e-10, the same support."""
    from oracle import cloops, entry, readme
    xb, yb, bt = readme.bp_data()
    rng = np.random.default_rng(8)
    A = rng.standard_normal((61, 233))
    b0 = np.zeros(233); b0[rng.choice(233, 9, replace=False)] = rng.standard_normal(9) * 3
    opts = dict(maxit=10000, eps_abs=1e-4, eps_rel=1e-4, rho_ratio=1.0)
    for x, y, N, nt in ((xb, yb, 2, 1), (xb, yb, 7, 3), (A, A @ b0, 3, 1), (A, A @ b0, 5, 3)):
        d, trc = {"trace": []}, []
        ref = entry.admm_parbp(x, y, N, opts, d)
        got = cloops.admm_parbp_c(x, y, N, opts, nthreads=nt, trace=trc)
        assert got["niter"] == ref["niter"], (N, got["niter"], ref["niter"])
        a, r = np.asarray(trc), np.asarray(d["trace"], dtype=np.float64)
        assert a.shape == r.shape
        assert np.array_equal(a[:, [0, 5, 6]], r[:, [0, 5, 6]])
        for col in range(1, 5):                                   # (the dual residual is a difference of sums: measured against the column's scale, as tests/test_gpu_parbp.py does)
            assert np.abs(a[:, col] - r[:, col]).max() < 1e-9 * np.abs(r[:, col]).max(), (N, col)
        assert relerr(got["beta"], ref["beta"]) < 1e-10 and np.array_equal(got["beta"] != 0, ref["beta"] != 0)
        assert abs(got["rho"] / d["solver"].rho - 1) < 1e-14
    assert np.abs(got["beta"] - b0).max() < 5e-3                  # (and it is basis pursuit: the sparse truth comes back)
