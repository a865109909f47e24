"""Dev tool (GPU box): single-launch tall iteration against the two-launch path -- bit-identical results, timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import numpy as np
from admm_amd import admm_lasso, DevicePtr, LassoPlan
from helpers import synth_lasso


def fit_env(model, val):
    os.environ["ADMM_HIP_TALL_FUSED"] = val
    plan = LassoPlan(model)
    plan.run()
    t0 = time.time(); fit = plan.run(); dt = time.time() - t0
    plan.close()
    return fit, dt

for (n, p, nl) in ((4700, 2300, 8), (9000, 4200, 10)):
    x, y = synth_lasso(n, p, 40, seed=p)
    a, ta = fit_env(admm_lasso(x, y).penalty(nlambda=nl), "0")
    b, tb = fit_env(admm_lasso(x, y).penalty(nlambda=nl), "1")
    same = np.array_equal(a.beta_dense, b.beta_dense) and list(a.niter) == list(b.niter)
    print(f"p={p}: variants {a.stats['xupdate_variant']} / {b.stats['xupdate_variant']}  identical={same}  niter {list(map(int, b.niter))}  "
          f"us/iter {a.stats['t_loop'] / a.stats['total_iter'] * 1e6:.1f} -> {b.stats['t_loop'] / b.stats['total_iter'] * 1e6:.1f}")
    assert same
# C2 size
dev = torch.device("cuda", 0)
n, p = 100000, 10000
g = torch.Generator(device=dev); g.manual_seed(123)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0 + 1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
bt = torch.zeros(p, dtype=torch.float64, device=dev); bt[:1000] = torch.rand(1000, generator=g, device=dev, dtype=torch.float64)
y = bt @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
res = {}
for val in ("0", "1", "2"):
    os.environ["ADMM_HIP_TALL_FUSED"] = val
    plan = LassoPlan(admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100))
    plan.run()
    ts = []
    for _ in range(3):
        fit = plan.run(); ts.append(fit.stats["t_loop"] / fit.stats["total_iter"] * 1e6)
    res[val] = fit
    print(f"C2 fused={val}: variant {fit.stats['xupdate_variant']} iterations {int(fit.stats['total_iter'])} us/iter {min(ts):.2f} (runs {['%.2f' % t for t in ts]})")
    plan.close()
print("C2 identical:", all(np.array_equal(res["0"].beta_dense, res[v].beta_dense) and list(res["0"].niter) == list(res[v].niter) for v in ("1", "2")))
