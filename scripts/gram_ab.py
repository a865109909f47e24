"""Dev tool (GPU box): Gram kernel A/B -- tile order square vs row -- on the C2 shape; prints t_gram of a warm plan create."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from admm_amd import admm_lasso, DevicePtr, LassoPlan
dev = torch.device("cuda", 0)
n, p = int(os.environ.get("GRAM_AB_N", 100000)), int(os.environ.get("GRAM_AB_P", 10000))
g = torch.Generator(device=dev); g.manual_seed(123)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0 + 1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
y = torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
for order in (sys.argv[1:] or ["row", "square", "row", "square"]):
    os.environ["ADMM_HIP_GRAM_ORDER"] = order
    plan = LassoPlan(admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=3))
    fit = plan.run()
    print(order, "t_gram %.4f t_factor %.4f" % (fit.stats["t_gram"], fit.stats["t_factor"]), flush=True)
    plan.close()
