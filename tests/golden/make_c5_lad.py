"""Writes tests/golden/c5_lad_fixed_maxit.npz: the oracle's admm_lad output on BASELINE configs[4] (n=50000, p=5000,
fp64) after a FIXED number of iterations, for the data `c5_lad_data(seed)` below generates (NumPy PCG64 stream: the test
regenerates the same arrays from the seed).  Data only: beta (p+1 doubles), niter, and the generator's parameters.

    python tests/golden/make_c5_lad.py        (about 2 minutes of CPU, ~6 GB of RAM)

Follows the README's LAD recipe at scale (README.md:299-304: dense beta* ~ U(0,1), heavy-tailed noise), intercept = FALSE,
rho = 1 (R/20_admm_lad.R:28-31)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

N, P, SEED, MAXIT = 50000, 5000, 505, 25


def c5_lad_data(seed=SEED, n=N, p=P):
    rng = np.random.default_rng(seed)
    x = np.empty((n, p), order="F")
    for j0 in range(0, p, 250):                       # column chunks: the stream is the same for any chunking of whole columns
        x[:, j0:j0 + 250] = rng.standard_normal((n, min(250, p - j0))) * 2.0
    b = rng.uniform(size=p)
    y = x @ b + rng.standard_t(3, size=n)
    return x, y


def main():
    from oracle import entry
    x, y = c5_lad_data()
    out = {}
    for maxit in (MAXIT,):
        ref = entry.admm_lad(x, y, False, dict(entry.LAD_OPTS, maxit=maxit))
        out[f"beta_maxit{maxit}"] = ref["beta"]
        out[f"niter_maxit{maxit}"] = np.int64(ref["niter"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_lad_fixed_maxit.npz")
    np.savez_compressed(path, n=N, p=P, seed=SEED, maxit=MAXIT, **out)
    print("wrote", path, os.path.getsize(path), "bytes; niter", int(ref["niter"]))


if __name__ == "__main__":
    main()
