"""The C2 tall Gram alone (n = 1e5, p = 1e4, device input, default fp16 x 2 split): two plan creations, for counter passes of rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from admm_amd import DevicePtr, admm_lasso
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(123)
n, p = 100000, 10000
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0 + 1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
y = torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
import admm_amd
for k, v in [a.split("=") for a in sys.argv[1:]]:
    admm_amd.options.set(**{k: v})
for rep in range(2):
    fit = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=2).opts(maxit=2).fit()
    print("t_gram ms", fit.stats["t_gram"] * 1e3)
