#!/usr/bin/env python
"""Dev tool (GPU box): K-fold cross-validation of the C2-shaped problem (device-resident inputs).  bench_cv.py [nfolds] [n] [p]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from admm_amd import DevicePtr, admm_lasso, load
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
p = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
chunk = max(1, (1 << 27) // n)
for c0 in range(0, p, chunk):
    c1 = min(p, c0 + chunk)
    xt[c0:c1] = torch.randn((c1 - c0, n), generator=g, device=dev, dtype=torch.float64) * 2
b = torch.zeros(p, dtype=torch.float64, device=dev); b[:1000] = torch.rand(1000, generator=g, device=dev, dtype=torch.float64)
y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
lib = load()
m = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100)
m.fit()
t0 = time.time(); cv = m.cv(nfolds=K); lib.admm_hip_device_synchronize(); t = time.time() - t0
t0 = time.time(); m.fit(); lib.admm_hip_device_synchronize(); t1 = time.time() - t0
print({"nfolds": K, "n": n, "p": p, "cv_s": round(t, 3), "one_fit_s": round(t1, 3), "idx_min": cv.idx_min, "idx_1se": cv.idx_1se,
       "lambda_min": cv.lambda_min, "cvm_min": float(cv.cvm[cv.idx_min]), "fold_iters": [int(v) for v in cv.fold_niter.sum(axis=1)]})
