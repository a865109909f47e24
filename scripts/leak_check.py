#!/usr/bin/env python
"""Dev tool (GPU box): repeated fits of every entry point; device memory in use must return to its starting level."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
import admm_amd

rng = np.random.default_rng(0)
def data(n, p):
    x = rng.standard_normal((n, p)) * 2
    b = np.zeros(p); b[:5] = rng.uniform(size=5)
    return np.asfortranarray(x), x @ b + rng.standard_normal(n)

def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20

xt, yt = data(3000, 400)
xw, yw = data(300, 2500)
xb = np.asfortranarray(rng.standard_normal((200, 700)))
bb = np.zeros(700); bb[:10] = 1.0
yb = xb @ bb
for rep in range(3):
    u0 = used()
    for _ in range(15):
        admm_amd.admm_lasso(xt, yt).penalty(nlambda=8).fit()
        admm_amd.admm_lasso(xw, yw).penalty(nlambda=8).fit()
        admm_amd.admm_enet(xt, yt).penalty(nlambda=5, alpha=0.5).fit()
        admm_amd.admm_lasso(xt, yt).penalty(nlambda=3).parallel(3).fit()
        admm_amd.admm_lad(xt[:, :60], yt).fit()
        admm_amd.admm_bp(xb, yb).fit()
        admm_amd.admm_lasso(xt, yt).penalty(nlambda=4).cv(nfolds=3)
        admm_amd.admm_lasso(xt, yt).penalty(nlambda=4).fit_responses(np.stack([yt, yt[::-1]], axis=1))
        admm_amd.admm_bp(xb, yb).parallel(4).fit()                                   # sharing basis pursuit (round 3)
        admm_amd.admm_dantzig(xt[:, :40], yt).penalty(nlambda=3, lambda_min_ratio=0.1).opts(maxit=300).fit()
        admm_amd.options.set(CV_DOWNDATE="1")
        admm_amd.admm_lasso(xt, yt).penalty(nlambda=4).cv(nfolds=3)                  # folds as down-dates
        admm_amd.options.set(CV_DOWNDATE=None)
        admm_amd.options.set(REFINE="1")
        admm_amd.admm_lasso(xt, yt).penalty(nlambda=3).fit()                         # refined x-update keeps a second p x p matrix
        admm_amd.options.set(REFINE=None)
    print("round", rep, "device MiB in use before/after", round(u0, 1), round(used(), 1), flush=True)
