"""Dev tool (GPU box): hipGraph replay of the tall loop's iteration batches vs plain launches (C2 shape, no sampling)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from admm_amd import admm_lasso, DevicePtr, LassoPlan
dev = torch.device("cuda", 0)
n, p = 100000, 10000
g = torch.Generator(device=dev); g.manual_seed(123)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0 + 1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
bt = torch.zeros(p, dtype=torch.float64, device=dev); bt[:1000] = torch.rand(1000, generator=g, device=dev, dtype=torch.float64)
y = bt @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
plan = LassoPlan(admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100))
res = {}
for val in ("0", "1", "0", "1"):
    os.environ["ADMM_HIP_TALL_GRAPH"] = val
    plan.run()
    ts = []
    for _ in range(3):
        fit = plan.run(); ts.append(fit.stats["t_loop"] / fit.stats["total_iter"] * 1e6)
    res[val] = fit
    print(f"graph={val}: iterations {int(fit.stats['total_iter'])} us/iter {min(ts):.2f} (runs {['%.2f' % t for t in ts]})", flush=True)
print("identical:", np.array_equal(res["0"].beta_dense, res["1"].beta_dense) and list(res["0"].niter) == list(res["1"].niter))
