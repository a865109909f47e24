"""What ONE GPU of an 8-GPU run of BASELINE configs[3] does per iteration: one 1250 x 100 000 Woodbury worker (K = 1 block in this process) with the
exchange over PEER (a one-rank communicator: the pushes and waits run, the link latency is absent).  Prints microseconds per iteration; the 8-GPU
projection is this figure + the xGMI hop against the one-GPU figure of bench.py's c4 line (8 blocks on one GPU)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from admm_amd import DevicePtr, load
from admm_amd import dist as adist
lib = load()
dev = torch.device("cuda", 0)
assert lib.admm_hip_set_device(0) == 0
adist.init_comm_peer(1, 0, lambda mine: mine)
n, p = 1250, 100000
g = torch.Generator(device=dev); g.manual_seed(123)
xt = torch.randn((p, n), generator=g, device=dev, dtype=torch.float64) * 2.0
b = torch.zeros(p, dtype=torch.float64, device=dev); b[:100] = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
plan = adist.DistLassoPlan(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n, p, 1, nlambda=3, lambda_min_ratio=0.3, n_local=n, maxit=1500)
plan.run()
fit = plan.run()
it, ls = int(fit.stats["total_iter"]), float(fit.stats["t_loop"])
print(json.dumps({"block": [n, p], "iterations": it, "loop_s": ls, "us_per_iter": ls / it * 1e6, "exchange_variant": int(fit.stats["exchange_variant"]),
                  "stream_bytes": 4.0 * n * p, "stream_us_at_6.6TBps": 4.0 * n * p / 6.6e12 * 1e6}))
plan.close()
adist.finalize_comm()
