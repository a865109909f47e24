#!/usr/bin/env python
"""Dev tool: timeline of the tall path's two launches per iteration from the in-kernel timestamps of a probe build
(ADMM_HIP_EXTRA_CXXFLAGS=-DADMM_HIP_PROBE python -m admm_amd.build --force; ADMM_HIP_PROBE_OUT=f python bench.py --steps 1 --warmup 0 ...).
Observers: 0 = tail workgroup 0, 1 = last tail workgroup, 2 = the decision workgroup riding the mat-vec launch."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(4096, 4, 8).astype(np.float64) * 0.01   # 100 MHz -> us
t0, t1, dc, ti = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
ok = (t0[:, 3] > 0) & (t1[:, 3] > 0) & (dc[:, 3] > 0)
idx = np.nonzero(ok)[0]
order = idx[np.argsort(dc[idx, 0])]
t0, t1, dc, ti = t0[order], t1[order], dc[order], ti[order]
nxt = np.roll(dc[:, 0], -1)
per = nxt - dc[:, 0]
act = (per > 0) & (per < 100)
print("records", len(order), "pairs", int(act.sum()), "median period us", np.median(per[act]))


def show(name, v):
    v = v[act]
    print(f"{name:52s} median {np.median(v):7.2f}  mean {v.mean():7.2f}  p90 {np.percentile(v, 90):7.2f}")


show("decision WG: entry -> control block back", dc[:, 1] - dc[:, 0])
show("decision WG: norm partials + block sum", dc[:, 2] - dc[:, 1])
show("decision WG: decision + publish", dc[:, 3] - dc[:, 2])
show("tail WG0 entry - decision WG entry (= mat-vec launch)", t0[:, 0] - dc[:, 0])
show("tail WG0: entry -> partial sums ready", t0[:, 1] - t0[:, 0])
show("tail WG0: element update", t0[:, 2] - t0[:, 1])
show("tail WG0: block sum + store", t0[:, 3] - t0[:, 2])
show("tail last WG entry - WG0 entry", t1[:, 0] - t0[:, 0])
show("tail last WG: entry -> end", t1[:, 3] - t1[:, 0])
show("next decision WG entry - tail last WG end", nxt - t1[:, 3])
show("next decision WG entry - tail WG0 entry", nxt - t0[:, 0])
show("period", per)
if ti[:, 1].max() > 0:
    for k, nm in enumerate(("first tile", "middle tile", "last tile", "tile at 3/4")):
        show(f"{nm}: entry - decision WG entry", ti[:, 2 * k] - dc[:, 0])
        show(f"{nm}: entry -> end", ti[:, 2 * k + 1] - ti[:, 2 * k])
    show("tail WG0 entry - last tile end", t0[:, 0] - ti[:, 5])
