#!/usr/bin/env python
"""Kernel durations of a rocprofv3 (rocpd sqlite) trace grouped by kernel name AND grid size (the same gemv kernel serves
very different shapes).  Usage: rocpd_by_grid.py results.db name-substring"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
gcol = [c for c in cols if "grid" in c.lower()]
sel = ", ".join(gcol[:3]) if gcol else "0"
rows = db.execute(f"select name, start, end, {sel} from kernels order by start").fetchall()
stats = {}
for r in rows:
    if sys.argv[2] not in r[0]:
        continue
    d = (r[2] - r[1]) / 1e3
    key = tuple(r[3:])
    st = stats.setdefault(key, [0, 0.0, 1e30, 0.0])
    st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
print("columns:", gcol)
for key, (c, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"grid {key}: calls {c} total {t/1e3:.2f} ms avg {t/c:.2f} us min {mn:.2f} max {mx:.2f}")
