"""Sum the counters of one kernel (substring match) in a rocprofv3 rocpd database.  Usage: pmc_kernel.py results.db <kernel substring>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
kn = "kernel_name" if "kernel_name" in cols else "name"
cn = "counter_name"; cv = "value" if "value" in cols else "counter_value"
agg = collections.defaultdict(lambda: [0, 0.0])
for name, c, v in db.execute(f"select {kn}, {cn}, {cv} from {view}"):
    if sys.argv[2] in name:
        a = agg[c]; a[0] += 1; a[1] += v
for c, (k, v) in sorted(agg.items()):
    print(f"{c:28s} launches {k:5d}  total {v:.4g}  per launch {v / k:.4g}")
