"""C3 shape (n = 2000, p = 200 000): the persistent stretch with 4 (default at this n), 2 and 1 column groups, on the first 40 lambdas of the
automatic grid (small active sets) and on the whole path: microseconds per iteration inside the stretch (option WIDE_PERSIST_STATS prints them)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import admm_amd
from admm_amd import DevicePtr, admm_lasso
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(123)
n, p = 2000, 200000
xt = torch.randn((p, n), generator=g, device=dev, dtype=torch.float64) * 2.0
b = torch.zeros(p, dtype=torch.float64, device=dev); b[:100] = torch.rand(100, generator=g, device=dev, dtype=torch.float64)
y = b @ xt + torch.randn(n, generator=g, device=dev, dtype=torch.float64)
torch.cuda.synchronize()
full = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(nlambda=100).fit()
lam = full.lambda_
for nl in (40, 100):
    for c in ("4", "2", "1"):
        with admm_amd.options(WIDE_ROWS_C=c, WIDE_PERSIST_STATS="1"):
            m = admm_lasso(DevicePtr(xt.data_ptr()), DevicePtr(y.data_ptr()), n=n, p=p).penalty(list(lam[:nl]))
            m.fit()
            f = m.fit()
        st = f.stats
        print(f"RESULT nl={nl} C={c}: {int(st['total_iter'])} iterations, {st['t_loop']*1e3:.1f} ms loop = {st['total_iter']/st['t_loop']:.0f} it/s, persist_iter {int(st['persist_iter'])}, nnz last {int(np.count_nonzero(f.beta_dense[1:, -1]))}", flush=True)
