#!/usr/bin/env python
"""Whole-$fit() wall times on the reference README's own performance cases (host inputs, like R would pass them),
to set beside the published numbers in BASELINE.md section 1 (unknown CPU).  Prints one JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from admm_amd import admm_bp, admm_enet, admm_lad, admm_lasso  # noqa: E402
from oracle.rrng import RRandom  # noqa: E402
from oracle import readme  # noqa: E402


def timed(f, reps=5):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.time()
        r = f()
        ts.append(time.time() - t0)
    return r, float(np.median(ts))


def lasso_case(n, p, m):
    r = RRandom(123)                                      # README.md:195-201 / 246-252
    b = np.concatenate([r.runif(m), np.zeros(p - m)])
    x = r.rnorm(n * p, sd=2.0).reshape((n, p), order="F")
    y = x @ b + r.rnorm(n)
    return np.asfortranarray(x), y


out = []
x, y = lasso_case(10000, 1000, 100)
fit, t = timed(lambda: admm_lasso(x, y).penalty(nlambda=100).fit())
out.append({"case": "lasso tall n=10000 p=1000, 100-lambda path", "gpu_ms": t * 1e3, "published_admm_ms": 321.0, "published_glmnet_ms": 1043.3,
            "iters": int(fit.niter.sum()), "loop_ms": fit.stats["t_loop"] * 1e3})
fit, t = timed(lambda: admm_lasso(x, y).penalty(nlambda=100).parallel().fit())
out.append({"case": "padmm[lasso] n=10000 p=1000 (2 row blocks)", "gpu_ms": t * 1e3, "published_admm_ms": 512.5, "iters": int(fit.niter.sum())})
fit, t = timed(lambda: admm_enet(x, y).penalty(nlambda=100, alpha=0.6).fit())
out.append({"case": "enet alpha=0.6 n=10000 p=1000", "gpu_ms": t * 1e3, "published_admm_ms": 289.0, "iters": int(fit.niter.sum())})
x, y = lasso_case(1000, 2000, 100)
fit, t = timed(lambda: admm_lasso(x, y).penalty(nlambda=100).fit())
out.append({"case": "lasso wide n=1000 p=2000, 100-lambda path", "gpu_ms": t * 1e3, "published_admm_ms": 247.4, "published_glmnet_ms": 199.4,
            "iters": int(fit.niter.sum())})
r = RRandom(123)                                           # README.md:296-304
n, p = 1000, 500
b = r.runif(p)
x = r.rnorm(n * p, sd=2.0).reshape((n, p), order="F")
y = x @ b + r.rnorm(n)
fit, t = timed(lambda: admm_lad(x, y, intercept=False).fit())
out.append({"case": "LAD n=1000 p=500", "gpu_ms": t * 1e3, "published_admm_ms": 51.6, "iters": fit.niter})
x, y, bt = readme.bp_data(1000, 2000, 100)
fit, t = timed(lambda: admm_bp(x, y).fit())
out.append({"case": "BP n=1000 p=2000", "gpu_ms": t * 1e3, "published_admm_ms": 292.0, "iters": fit.niter})
for o in out:
    print(json.dumps(o), flush=True)
