"""Writes tests/golden/c2_short_path.npz: the COMPILED C oracle (oracle/c/admm_tall_cpu.c, mode 0 = the reference's
arithmetic: float Cholesky factor, two triangular solves per iteration, one thread) on the headline problem width
p = 10 000 -- a short warm-started lambda path with a small fixed `maxit`, for data `c2_short_data(seed)` below generates
(NumPy PCG64 stream: the test regenerates the same arrays from the seed).  Data only: the lambdas, beta ((p+1) x nlambda
float32), niter, rho, and the oracle's decision trace (one row per iteration: lambda index, iteration, eps_p, eps_d,
r_p, r_d, c, outcome) so that a test can tell a different stopping / restart decision from a different iterate.

    python tests/golden/make_c2_short.py        (about 3 minutes on 8 cores, ~5 GB of RAM)

BASELINE configs[1] recipe (X ~ N(0, 2^2), beta* = U(0,1) on the first m = 1000, unit noise,
standardize = intercept = TRUE, eps 1e-5) with n cut to 2 p so that the NumPy Gram of the setup stays in minutes; the
lambdas are the first ten of the automatic 100-lambda grid (Lasso.cpp:78-89): the path goes from the null model to ~430
active coefficients exactly as the headline run starts; maxit = 42 lets the first lambdas stop on the rule (21 .. 40
iterations) and makes the later ones run out (niter = maxit + 1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N, P, M, SEED, MAXIT = 20000, 10000, 1000, 2024, int(os.environ.get("C2S_MAXIT", 42))
GRID_PICK = tuple(int(v) for v in os.environ.get("C2S_PICK", "0,1,2,3,4,5,6,7,8,9").split(","))


def c2_short_data(seed=SEED, n=N, p=P, m=M):
    rng = np.random.default_rng(seed)
    x = np.empty((n, p), order="F")
    for j0 in range(0, p, 500):                       # column chunks: the stream is the same for any chunking of whole columns
        x[:, j0:j0 + 500] = rng.standard_normal((n, min(500, p - j0))) * 2.0
    b = np.concatenate([rng.uniform(size=m), np.zeros(p - m)])
    y = x @ b + rng.standard_normal(n)
    return x, y


def main():
    import time
    from oracle import ctall, entry
    t0 = time.time()
    x, y = c2_short_data()
    print(f"data {time.time() - t0:.1f} s", flush=True)
    opts = dict(entry.LASSO_OPTS, maxit=1)
    grid = ctall.admm_lasso_c(x, y, None, 100, 1e-4, True, True, opts)["lambda"]      # the automatic grid (one iteration per lambda)
    lam = grid[list(GRID_PICK)]
    print(f"grid {time.time() - t0:.1f} s", flush=True)
    trace = []
    ref = ctall.admm_lasso_c(x, y, lam, 100, 1e-4, True, True, dict(entry.LASSO_OPTS, maxit=MAXIT), mode=0, nthreads=1, trace=trace)
    tr = np.asarray(trace)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c2_short_path.npz")
    np.savez_compressed(path, n=N, p=P, m=M, seed=SEED, maxit=MAXIT, lam=lam, beta=ref["beta"].astype(np.float32),
                        niter=ref["niter"].astype(np.int64), rho=np.float64(ref["rho"]), trace=tr)
    print("wrote", path, os.path.getsize(path), "bytes; niter", ref["niter"].tolist(), "rho", ref["rho"],
          "nnz", (ref["beta"][1:] != 0).sum(axis=0).tolist(), f"loop {ref['loop_seconds']:.1f} s, total {time.time() - t0:.1f} s")
    nc = tr[tr[:, 7] > 0]
    print("closest restart decision: c / (0.999 c_old) margins need the previous c; outcomes", np.bincount(tr[:, 7].astype(int)).tolist())
    gp = np.minimum(np.abs(tr[:, 4] - tr[:, 2]) / tr[:, 2], np.abs(tr[:, 5] - tr[:, 3]) / tr[:, 3])
    print("closest stopping test (relative distance of a residual from its threshold):", float(gp.min()))


if __name__ == "__main__":
    main()
