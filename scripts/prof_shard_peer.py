"""Dev tool (GPU box): the row-sharded tall solver with ONE rank over the PEER / RCCL exchange, for rocprofv3 --kernel-trace --stats.
    python scripts/prof_shard_peer.py <peer|rccl> [p] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from admm_amd import DevicePtr, load
from admm_amd import dist as adist
backend = sys.argv[1] if len(sys.argv) > 1 else "peer"
p = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
lib = load(); lib.admm_hip_set_device(0)
if backend == "peer":
    adist.init_comm_peer(1, 0, lambda mine: mine)
else:
    adist.init_comm(1, 0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
xt = torch.randn((p, n), generator=g, device=dev, dtype=torch.float64) * 2
b = torch.zeros(p, dtype=torch.float64, device=dev); b[:p // 10] = torch.rand(p // 10, generator=g, device=dev, dtype=torch.float64)
y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
plan = adist.DistLassoPlan(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n, p, 0, nlambda=30, lambda_min_ratio=1e-3, n_local=n)
plan.run()
t0 = time.time(); fit = plan.run(); dt = time.time() - t0
it = int(fit.stats["total_iter"])
print(backend, "iterations", it, "us/iter", fit.stats["t_loop"] / it * 1e6)
plan.close(); adist.finalize_comm()
