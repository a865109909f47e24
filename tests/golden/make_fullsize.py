"""Writes the fixed-maxit ORACLE fixtures at the BASELINE shapes that had none (round 5; C2 and C5-LAD have theirs):

    c3_fixed_maxit.npz     admm_lasso, wide: n = 2000, p = 200 000 (configs[2]); 3 lambdas of the automatic 100-grid x maxit 40
                           (regular steps at counters 0 / 3 / 15 and the active-set steps between them, ADMMLassoWide.h:121-155)
    c4_fixed_maxit.npz     admm_lasso$parallel(8): n = 10 000, p = 100 000 (configs[3]), 8 row blocks of 1250 x 10^5 -- the Woodbury
                           branch of PADMMLasso.h:23-30; lambda = 0.3 and 0.25 lambda_max x maxit 600 (the consensus iteration with
                           rho = lambda / K keeps z = 0 for its first ~500 iterations: a 25-iteration fixture would never see a
                           non-zero consensus variable)
    c5_bp_fixed_maxit.npz  admm_bp: n = 5000, p = 50 000 fp64 (configs[4]), maxit 25

each from oracle/entry.py (the NumPy restatement of the reference, pinned on the README vectors) on data the generators below
make from a seed (NumPy PCG64 stream, whole-column chunks: the tests regenerate the same arrays).  Data only: the lambdas, the
coefficients, niter, rho / the loose spectral radius, and the oracle's decision trace (so that a test can tell a different
decision from a different iterate).

    python tests/golden/make_fullsize.py [c3] [c4] [bp]        (minutes of CPU each; c4 needs ~20 GB of RAM)

Recipes: SURVEY section 8(d) / README.md:195-201, 370-377 (X ~ N(0, 2^2), beta* = U(0,1) on the first 100, unit noise,
standardize = intercept = TRUE, eps 1e-5; BP: A ~ N(0,1), 500 non-zeros U(0,1) at random positions, y = A beta* exactly, eps 1e-4)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

C3 = dict(n=2000, p=200000, m=100, seed=3003, maxit=40, pick=(5, 25, 50))
C4 = dict(n=10000, p=100000, m=100, seed=4004, maxit=600, K=8, fractions=(0.3, 0.25))
BP = dict(n=5000, p=50000, m=500, seed=5005, maxit=25, scale=20.0)


def lasso_data(seed, n, p, m):
    rng = np.random.default_rng(seed)
    x = np.empty((n, p), order="F")
    step = max(1, (1 << 24) // n)
    for j0 in range(0, p, step):                      # whole columns per chunk: the stream is the same for any chunking
        x[:, j0:j0 + step] = rng.standard_normal((n, min(step, p - j0))) * 2.0
    b = rng.uniform(size=m)
    y = x[:, :m] @ b + rng.standard_normal(n)
    return x, y


def bp_data(seed, n, p, m, scale=1.0):
    """`scale` multiplies beta*: with the README's U(0,1) coefficients and rho = 1 the minimum-norm start A'(AA')^-1 b stays below the
    soft threshold 1 / rho, z = 0 for the first iterations, the iterate REPEATS and the restart test c < 0.999 c_old (FADMMBase.h:243) is
    an exact tie up to the order of a sum, several times in a row (oracle/stepcheck.py, rounding_ties) -- two executions of the
    reference's own arithmetic then part ways at iteration 2 and no fixture can hold both.  U(0, 20) coefficients start with z != 0."""
    rng = np.random.default_rng(seed)
    a = np.empty((n, p), order="F")
    step = max(1, (1 << 24) // n)
    for j0 in range(0, p, step):
        a[:, j0:j0 + step] = rng.standard_normal((n, min(step, p - j0)))
    bt = np.zeros(p)
    bt[rng.choice(p, m, replace=False)] = rng.uniform(size=m) * scale
    return a, a @ bt, bt


def make_c3():
    from oracle import entry
    c = C3
    t0 = time.time()
    x, y = lasso_data(c["seed"], c["n"], c["p"], c["m"])
    det = {"trace": []}
    opts = dict(entry.LASSO_OPTS, maxit=c["maxit"])
    # the picked lambdas of the automatic 100-grid (Lasso.cpp:78-89) need lambda_0 = max|X'y| of the standardised float data (ADMMLassoWide.h:197)
    from oracle.datastd import DataStd
    dx, dy = np.array(x, dtype=np.float32, order="F"), np.array(y, dtype=np.float32)
    std = DataStd(c["n"], c["p"], True, True, np.float32)
    std.standardize(dx, dy)
    lambda0 = np.float32(np.abs((dx.T @ dy).astype(np.float32)).max())
    lam = entry._lambda_grid(lambda0, c["n"], std.scaleY, 100, 0.01)[list(c["pick"])]
    del dx, dy
    ref = entry.admm_lasso(x, y, lam, 100, 0.01, True, True, opts, detail=det)
    s = det["solver"]
    path = os.path.join(HERE, "c3_fixed_maxit.npz")
    np.savez_compressed(path, **{k: v for k, v in c.items() if k != "pick"}, pick=np.asarray(c["pick"]), lam=lam, beta=ref["beta"].astype(np.float32),
                        niter=ref["niter"].astype(np.int64), sprad=np.float64(s.sprad), lambda0=np.float64(s.lambda0), trace=np.asarray(det["trace"], dtype=np.float64))
    print("wrote", path, os.path.getsize(path), "bytes; niter", ref["niter"].tolist(), "sprad", float(s.sprad), "nnz", (ref["beta"][1:] != 0).sum(axis=0).tolist(),
          f"{time.time() - t0:.0f} s", flush=True)


def make_c4():
    from oracle import entry
    c = C4
    t0 = time.time()
    x, y = lasso_data(c["seed"], c["n"], c["p"], c["m"])
    det = {"trace": []}
    from oracle.datastd import DataStd
    dx, dy = np.array(x, dtype=np.float32, order="F"), np.array(y, dtype=np.float32)
    std = DataStd(c["n"], c["p"], True, True, np.float32)
    std.standardize(dx, dy)
    lambda0 = np.float64(np.float32(np.abs((dx.T @ dy).astype(np.float32)).max()))            # PADMMLasso.h:161
    lmax = lambda0 / c["n"] * np.float64(std.scaleY)                                          # Lasso.cpp:82
    del dx, dy
    lam = np.asarray([f * lmax for f in c["fractions"]])
    ref = entry.admm_parlasso(x, y, lam, 2, 0.5, True, True, c["K"], dict(entry.LASSO_OPTS, maxit=c["maxit"]), detail=det)
    s = det["solver"]
    # how far the reference's OWN arithmetic is from itself at these unconverged iterates: the same run with the workers' systems solved
    # exactly (oracle/variants.py "exact") instead of by the float LLT.  z has only just left zero (soft threshold of a value within
    # 1e-3 of its threshold), so a 1e-7 rounding of x shows as 1e-4 .. 1e-3 of the column: the test allows 5 x this drift (rule R3 of
    # tests/helpers.py), not a flat 1e-4
    from oracle.solvers import PADMMLasso
    PADMMLasso.xmode = "exact"
    try:
        alt = entry.admm_parlasso(x, y, lam, 2, 0.5, True, True, c["K"], dict(entry.LASSO_OPTS, maxit=c["maxit"]))
    finally:
        PADMMLasso.xmode = "llt32"
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import col_err                                                               # the tests' per-column metric
    floor = 1e-2 * float(np.abs(ref["beta"]).max())
    drift = [float(col_err(alt["beta"][:, j], ref["beta"][:, j], floor)) for j in range(len(lam))]
    print("drift of the exact-solve variant per column:", drift, flush=True)
    path = os.path.join(HERE, "c4_fixed_maxit.npz")
    c = {k: v for k, v in c.items() if k != "fractions"}
    np.savez_compressed(path, **c, fractions=np.asarray(C4["fractions"]), drift=np.asarray(drift), lam=ref["lambda"], beta=ref["beta"].astype(np.float32), niter=ref["niter"].astype(np.int64), rho=np.float64(s.rho),
                        trace=np.asarray(det["trace"], dtype=np.float64))
    print("wrote", path, os.path.getsize(path), "bytes; niter", ref["niter"].tolist(), "rho", float(s.rho), "nnz", (ref["beta"][1:] != 0).sum(axis=0).tolist(),
          f"{time.time() - t0:.0f} s", flush=True)


def make_bp():
    from oracle import entry
    c = BP
    t0 = time.time()
    a, b, _ = bp_data(c["seed"], c["n"], c["p"], c["m"], c["scale"])
    det = {"trace": []}
    ref = entry.admm_bp(a, b, dict(entry.BP_OPTS, maxit=c["maxit"]), detail=det)
    s = det["solver"]
    path = os.path.join(HERE, "c5_bp_fixed_maxit.npz")
    np.savez_compressed(path, **c, beta=ref["beta"], niter=np.int64(ref["niter"]), rho=np.float64(s.rho), trace=np.asarray(det["trace"], dtype=np.float64))
    t = np.asarray(det["trace"], dtype=np.float64)
    nc = t[t[:, 8] > 0]
    print("closest restart test |c / (0.999 c_old) - 1|:", float(np.abs(nc[:, 6] / (0.999 * nc[:, 7]) - 1).min()), flush=True)
    print("wrote", path, os.path.getsize(path), "bytes; niter", int(ref["niter"]), "final rho", float(s.rho), "nnz", int(np.count_nonzero(ref["beta"])),
          f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4", "bp"]
    for w in which:
        {"c3": make_c3, "c4": make_c4, "bp": make_bp}[w]()
