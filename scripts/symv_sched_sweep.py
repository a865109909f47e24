"""Dev tool (GPU box): the headline workload (C2: n = 100000, p = 10000, 100-lambda warm-started path) under different
segment schedules of the symmetric x-update (ADMM_HIP_SYMV_SCHED = width_big,width_small,split_permille, symv_kernels.h):
iterations/s of the whole loop, average x-update launch (HIP events on sampled launches), and the largest coefficient
difference against the first schedule (a different partial grouping rounds differently; it must stay at rounding level).

    python scripts/symv_sched_sweep.py [p] [n] "128,128,0" "192,64,550" ...        -> one JSON line per schedule
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import numpy as np  # noqa: E402

args = sys.argv[1:]
p = int(args.pop(0)) if args and args[0].isdigit() else 10000
n = int(args.pop(0)) if args and args[0].isdigit() else 100000
scheds = args or ["128,128,0", "192,64,550"]
os.environ["ADMM_HIP_PROFILE_STRIDE"] = "32"
import admm_amd  # noqa: E402
from admm_amd import admm_lasso, DevicePtr, LassoPlan, load  # noqa: E402
lib = load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(123)
xt = torch.empty((p, n), dtype=torch.float64, device=dev)
chunk = max(1, (1 << 27) // n)
for c0 in range(0, p, chunk):
    c1 = min(p, c0 + chunk)
    xt[c0:c1] = torch.randn((c1 - c0, n), generator=g, device=dev, dtype=torch.float64) * 2.0
bt = torch.zeros(p, dtype=torch.float64, device=dev)
bt[:p // 10] = torch.rand(p // 10, generator=g, device=dev, dtype=torch.float64)
y = bt @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
model = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100)
ref = None
for sc in scheds:
    admm_amd.options.set(SYMV_SCHED=sc)
    plan = LassoPlan(model)
    fit = plan.run()
    best = None
    for rep in range(4):
        lib.admm_hip_device_synchronize()
        t0 = time.time()
        fit = plan.run()
        lib.admm_hip_device_synchronize()
        dt = time.time() - t0
        it = int(fit.stats["total_iter"])
        rec = dict(sched=sc, iters=it, it_per_s=it / dt, us_per_iter=1e6 * dt / it, loop_ms_events=fit.stats["loop_ms_events"],
                   xupdate_us=1e3 * fit.stats["xupdate_ms_avg"], variant=int(fit.stats["xupdate_variant"]))
        if best is None or rec["it_per_s"] > best["it_per_s"]:
            best = rec
    b = fit.beta_dense.astype(np.float64)
    if ref is None:
        ref = b
    best["max_beta_diff_vs_first"] = float(np.abs(b - ref).max() / np.abs(ref).max())
    best["niter_sum"] = int(fit.niter.sum())
    print(json.dumps(best), flush=True)
    plan.close()
