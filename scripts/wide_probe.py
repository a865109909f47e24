#!/usr/bin/env python
"""Dev tool: timeline of the wide path's two launches per iteration from the in-kernel timestamps of a probe build
(ADMM_HIP_EXTRA_CXXFLAGS=-DADMM_HIP_PROBE python -m admm_amd.build --force; ADMM_HIP_PROBE_OUT=f python scripts/bench_configs.py c3).
Usage: wide_probe.py f"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(4096, 4, 8).astype(np.float64) * 0.01   # 100 MHz -> us
x0, x1, x2, tl = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
ok = (x0[:, 5] > 0) & (tl[:, 2] > 0) & (x1[:, 5] > 0)
idx = np.nonzero(ok)[0]
# order records by time
order = idx[np.argsort(x0[idx, 0])]
x0, x1, x2, tl = x0[order], x1[order], x2[order], tl[order]
nxt = np.roll(x0[:, 0], -1)
per = nxt - x0[:, 0]
act = (per > 0) & (per < 40)          # consecutive active-set iterations
print("records", len(order), "active pairs", int(act.sum()), "median period us", np.median(per[act]))


def show(name, v):
    v = v[act]
    print(f"{name:46s} median {np.median(v):7.2f}  mean {v.mean():7.2f}  p90 {np.percentile(v, 90):7.2f}")


show("x WG0: entry -> prologue loads back", x0[:, 1] - x0[:, 0])
show("x WG0: decision", x0[:, 2] - x0[:, 1])
show("x WG0: stage t + barrier", x0[:, 3] - x0[:, 2])
show("x WG0: columns", x0[:, 4] - x0[:, 3])
show("x WG0: reduce waves + write partial", x0[:, 5] - x0[:, 4])
show("x WG255 entry - WG0 entry", x1[:, 0] - x0[:, 0])
show("x WG255: entry -> end", x1[:, 5] - x1[:, 0])
show("x WG255: columns", x1[:, 4] - x1[:, 3])
show("x last WG entry - WG0 entry", x2[:, 0] - x0[:, 0])
show("x last WG: entry -> decision done", x2[:, 2] - x2[:, 0])
show("x end (max of observers) - WG0 entry", np.maximum(np.maximum(x0[:, 5], x1[:, 5]), x2[:, 2]) - x0[:, 0])
show("tail WG0 entry - x observers' end", tl[:, 0] - np.maximum(np.maximum(x0[:, 5], x1[:, 5]), x2[:, 2]))
show("tail WG0: entry -> loads back", tl[:, 1] - tl[:, 0])
show("tail WG0: compute + block sum + store", tl[:, 2] - tl[:, 1])
show("next x WG0 entry - tail WG0 end", nxt - tl[:, 2])
show("period", per)
