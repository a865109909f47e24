"""Dev tool (GPU box): repeated fits through every entry point must not grow device memory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from admm_amd import admm_lasso, admm_enet, admm_lad, admm_bp, LassoPlan
rng = np.random.default_rng(0)
xt, yt = rng.standard_normal((600, 300)), rng.standard_normal(600)
xw, yw = rng.standard_normal((100, 400)), rng.standard_normal(100)
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
base = None
for rep in range(40):
    admm_lasso(xt, yt).penalty(nlambda=5).fit()
    admm_enet(xw, yw).penalty(nlambda=5, alpha=0.5).fit()
    m = admm_lasso(xt, yt).penalty(nlambda=3).opts(maxit=50); m.nthread = 3; m.fit()
    admm_lad(xt, yt).opts(maxit=50).fit()
    admm_bp(xw, yw).opts(maxit=50).fit()
    p = LassoPlan(admm_lasso(xt, yt).penalty(nlambda=4)); p.run(); p.run(); p.close()
    if rep in (4, 39):
        print("rep", rep, "device MiB in use", round(used(), 1), flush=True)
        if rep == 4: base = used()
assert used() - base < 64, (used(), base)
print("no growth")
