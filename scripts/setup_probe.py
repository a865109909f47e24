"""Dev tool (GPU box): where does the one-time setup of the C2 tall problem spend its time, call after call?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from admm_amd import admm_lasso, DevicePtr, LassoPlan, load
lib = load()
dev = torch.device("cuda", 0)
n, p = 100000, 10000
g = torch.Generator(device=dev); g.manual_seed(123)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0 + 1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
y = torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
model = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=3)
for k in range(5):
    t0 = time.time()
    plan = LassoPlan(model)
    lib.admm_hip_device_synchronize()
    t1 = time.time()
    fit = plan.run()
    t2 = time.time()
    plan.close()
    lib.admm_hip_device_synchronize()
    t3 = time.time()
    s = fit.stats
    print(f"create {t1 - t0:.4f} (std {s['t_standardize']:.4f} gram {s['t_gram']:.4f} eigs {s['t_eigs']:.4f} factor {s['t_factor']:.4f} "
          f"other {t1 - t0 - s['t_standardize'] - s['t_gram'] - s['t_eigs'] - s['t_factor']:.4f}) run {t2 - t1:.4f} close {t3 - t2:.4f}", flush=True)
