"""Dev tool: duration histogram of one kernel from a rocprofv3 rocpd database.  rocpd_hist.py results.db name_substring [split_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2]; split = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
rows = [(e - s) / 1e3 for n, s, e in db.execute("select name, start, end from kernels order by start") if sub in n]
lo = [d for d in rows if d < split]; hi = [d for d in rows if d >= split]
import statistics as st
print(f"{sub}: {len(rows)} launches; < {split} us: n={len(lo)} avg={st.mean(lo):.2f} median={st.median(lo):.2f} p90={sorted(lo)[int(0.9*len(lo))]:.2f}; >= {split} us: n={len(hi)} avg={(st.mean(hi) if hi else 0):.1f}")
