import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
os.environ.setdefault("ADMM_HIP_PROFILE_STRIDE", sys.argv[1] if len(sys.argv) > 1 else "8")
from admm_amd import admm_lasso, DevicePtr, LassoPlan, load
lib = load(); lib.admm_hip_set_device(0)
dev = torch.device("cuda", 0)
n, p = 100000, 10000
g = torch.Generator(device=dev); g.manual_seed(1)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
for c0 in range(0, p, 1000):
    xt[c0:c0+1000] = torch.randn((1000, n), generator=g, device=dev, dtype=torch.float64) * 2.0
bt = torch.zeros(p, dtype=torch.float64, device=dev); bt[:100] = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
y = bt @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
plan = LassoPlan(admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100))
del xt; torch.cuda.empty_cache()
plan.run()
for i in range(5):
    t0 = time.time(); fit = plan.run(); w = time.time() - t0
    s = fit.stats
    print(f"stride={os.environ['ADMM_HIP_PROFILE_STRIDE']} layout={os.environ.get('ADMM_HIP_SYMV_LAYOUT','packed')} wall {w*1e3:.2f} ms  t_total {s['t_total']*1e3:.2f}  t_loop(wall) {s['t_loop']*1e3:.2f}  loop_events {s['loop_ms_events']:.2f}  iters {s['total_iter']} launches {s['xupdate_launches']} xms {s['xupdate_ms_avg']*1e3:.2f}us")
