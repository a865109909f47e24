#!/usr/bin/env python
"""Dev tool: phases of par_z_kernel from a probe build (see probe.h).  Observers: workgroup 0, last, middle."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(4096, 4, 8).astype(np.float64) * 0.01
for o, nm in enumerate(("WG 0", "last WG", "middle WG")):
    t = a[:, o]
    ok = t[:, 3] > 0
    t = t[ok]
    print(nm, "records", len(t), "| entry->decision %.2f | elements %.2f | block sum + store %.2f | total %.2f us (medians)" % (
        np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(t[:, 3] - t[:, 0])))
for o, nm in enumerate(("WG 0", "last WG", "middle WG")):
    t = a[:, o]
    t = t[t[:, 3] > 0]
    if t[:, 6].max() > 0:
        print(nm, "entry->ctl %.2f | nsum loads %.2f | decision math %.2f | publish %.2f" % (
            np.median(t[:, 4] - t[:, 0]), np.median(t[:, 5] - t[:, 4]), np.median(t[:, 6] - t[:, 5]), np.median(t[:, 1] - t[:, 6])))
t0, t1 = a[:, 0], a[:, 1]
ok = (t0[:, 3] > 0) & (t1[:, 3] > 0)
print("last WG entry - WG0 entry %.2f | last WG end - WG0 entry %.2f" % (np.median(t1[ok, 0] - t0[ok, 0]), np.median(t1[ok, 3] - t0[ok, 0])))
