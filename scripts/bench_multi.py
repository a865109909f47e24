#!/usr/bin/env python
"""Dev tool (GPU box): m responses of the C2-shaped problem through admm_hip_lasso_multi against m separate fits.
Usage: bench_multi.py [m] [n] [p]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import numpy as np  # noqa: E402
from admm_amd import DevicePtr, admm_lasso, load  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
chunk = max(1, (1 << 27) // n)
for c0 in range(0, p, chunk):
    c1 = min(p, c0 + chunk)
    xt[c0:c1] = torch.randn((c1 - c0, n), generator=g, device=dev, dtype=torch.float64) * 2
B = torch.zeros((m, p), dtype=torch.float64, device=dev)
B[:, :1000] = torch.rand((m, 1000), generator=g, device=dev, dtype=torch.float64)
Yt = B @ xt + torch.randn((m, n), generator=g, device=dev, dtype=torch.float64)      # m x n row-major == n x m column-major
torch.cuda.synchronize()
lib = load()
model = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(Yt.data_ptr()), n=n, p=p).penalty(nlambda=100)
model.fit_responses(DevicePtr(Yt.data_ptr()), m=1)                                      # warm-up (code objects, allocator)
t0 = time.time(); fits = model.fit_responses(DevicePtr(Yt.data_ptr()), m=m); lib.admm_hip_device_synchronize(); t_multi = time.time() - t0
t0 = time.time()
singles = []
for j in range(m):
    singles.append(admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(Yt[j].data_ptr()), n=n, p=p).penalty(nlambda=100).fit())
lib.admm_hip_device_synchronize(); t_single = time.time() - t0
same = all(np.array_equal(a.beta_dense, b.beta_dense) and np.array_equal(a.niter, b.niter) for a, b in zip(fits, singles))
print({"m": m, "n": n, "p": p, "multi_s": round(t_multi, 4), "separate_s": round(t_single, 4), "speedup": round(t_single / t_multi, 3),
       "bit_identical": same, "iters": [int(f.niter.sum()) for f in fits]})
