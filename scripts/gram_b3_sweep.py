#!/usr/bin/env python
"""t_gram of the C2 setup (n = 1e5, p = 1e4, device input) for the variants of the bf16 three-way-split Gram: K tiles per launch."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import admm_amd
from admm_amd import DevicePtr, admm_lasso
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(123)
n, p = 100000, 10000
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0 + 1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
b = torch.zeros(p, dtype=torch.float64, device=dev); b[:1000] = torch.rand(1000, generator=g, device=dev, dtype=torch.float64)
y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
variants = [("fp32", {"ADMM_HIP_GRAM_SPLIT": "0"})] + [(f"{m} ktiles={k}", {"ADMM_HIP_GRAM_SPLIT": m, "ADMM_HIP_GRAM_B3_KTILES": str(k)}) for m in ("bf16x3", "f16x2") for k in sys.argv[1:] or ["0", "516"]]
for name, env in variants:
    admm_amd.options.reset()
    admm_amd.options.set(**{k.replace("ADMM_HIP_", ""): v for k, v in env.items()})
    ts = []
    for rep in range(3):
        fit = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=2).opts(maxit=3).fit()
        ts.append(fit.stats["t_gram"])
    print(json.dumps({"variant": name, "t_gram_ms": [round(t * 1e3, 2) for t in ts], "rho": fit.stats["rho"]}), flush=True)
