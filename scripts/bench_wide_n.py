#!/usr/bin/env python
"""Dev tool (GPU box): a wide Lasso path at a given shape (n > 4096 takes the three-launch path).  bench_wide_n.py n p [nlambda]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from admm_amd import DevicePtr, admm_lasso
n, p = int(sys.argv[1]), int(sys.argv[2])
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
xt = torch.randn((p, n), generator=g, device=dev, dtype=torch.float64) * 2
b = torch.zeros(p, dtype=torch.float64, device=dev); b[:100] = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
m = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=nl)
m.fit()
fit = m.fit()
st = fit.stats
it = int(st["total_iter"])
print({"n": n, "p": p, "iterations": it, "loop_s": round(st["t_loop"], 4), "us_per_iter": round(st["t_loop"] / it * 1e6, 2),
       "nnz_last": int(np.count_nonzero(fit.beta_dense[1:, -1]))})
