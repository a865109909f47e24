"""Dev tool (GPU box): LAD / BP at sizes with several 128-blocks (fp64 matrix-core Gram, blocked Cholesky / inverse)
against the oracle.   python tests/tools/fuzz_dense.py [ncases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
from admm_amd import admm_lad, admm_bp
from oracle import entry

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
for c in range(ncases):
    kind = rng.choice(["lad", "bp"])
    maxit = int(rng.choice([30, 400]))
    if kind == "lad":
        p = int(rng.choice([129, 257, 300, 513, 640])); n = p + int(rng.integers(50, 1500)); icpt = bool(rng.integers(2))
        x = rng.standard_normal((n, p)); y = x[:, :5] @ rng.uniform(size=5) + rng.standard_t(2, size=n)
        t0 = time.time(); fit = admm_lad(x, y, icpt).opts(maxit=maxit).fit(); t1 = time.time()
        ref = entry.admm_lad(x, y, icpt, dict(entry.LAD_OPTS, maxit=maxit)); bg = np.asarray(fit.beta)
    else:
        n = int(rng.choice([129, 200, 257, 384, 500])); p = n + int(rng.integers(100, 2500))
        x = rng.standard_normal((n, p)); b = np.zeros(p); b[rng.choice(p, 10, replace=False)] = rng.standard_normal(10); y = x @ b
        t0 = time.time(); fit = admm_bp(x, y).opts(maxit=maxit).fit(); t1 = time.time()
        ref = entry.admm_bp(x, y, dict(entry.BP_OPTS, maxit=maxit)); bg = fit.beta.toarray().ravel()
    e = np.abs(bg - ref["beta"]).max() / max(np.abs(ref["beta"]).max(), 1e-300)
    dn = abs(int(fit.niter) - int(ref["niter"]))
    print(f"{c:2d} {kind:3s} n={n:5d} p={p:5d} maxit={maxit:4d} relerr={e:.2e} dniter={dn} (of {int(ref['niter'])}) gpu {t1 - t0:.2f}s {'SUSPECT' if (e > 1e-6 and dn == 0) or not np.isfinite(e) else ''}", flush=True)
