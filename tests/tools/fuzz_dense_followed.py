"""Dev tool (GPU box): the LAD / BP cases of fuzz_dense.py under the strict trace-based rule (assert_dense_followed).
python tests/tools/fuzz_dense_followed.py [ncases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import numpy as np
from admm_amd import admm_lad, admm_bp
from oracle import entry
from helpers import assert_dense_followed

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for c in range(ncases):
    kind = rng.choice(["lad", "bp"])
    maxit = int(rng.choice([30, 400]))
    try:
        if kind == "lad":
            p = int(rng.choice([129, 257, 300, 513, 640])); n = p + int(rng.integers(50, 1500)); icpt = bool(rng.integers(2))
            x = rng.standard_normal((n, p)); y = x[:, :5] @ rng.uniform(size=5) + rng.standard_t(2, size=n)
            fit = admm_lad(x, y, icpt).opts(maxit=maxit).fit(trace=True)
            assert_dense_followed("lad", np.asarray(fit.beta), fit.niter, fit.trace, x, y, dict(entry.LAD_OPTS, maxit=maxit), intercept=icpt, tol=1e-6,
                                  label=f"{c} lad n={n} p={p} maxit={maxit}")
        else:
            n = int(rng.choice([129, 200, 257, 384, 500])); p = n + int(rng.integers(100, 2500))
            x = rng.standard_normal((n, p)); b = np.zeros(p); b[rng.choice(p, 10, replace=False)] = rng.standard_normal(10); y = x @ b
            fit = admm_bp(x, y).opts(maxit=maxit).fit(trace=True)
            assert_dense_followed("bp", fit.beta.toarray().ravel(), fit.niter, fit.trace, x, y, dict(entry.BP_OPTS, maxit=maxit), tol=1e-6,
                                  label=f"{c} bp n={n} p={p} maxit={maxit}")
    except Exception as e:                                  # noqa: BLE001
        bad += 1
        print("FAIL case", c, kind, type(e).__name__, str(e)[:300], flush=True)
print("dense cases", ncases, "failures", bad)
