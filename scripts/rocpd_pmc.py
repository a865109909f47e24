#!/usr/bin/env python
"""Per-kernel PMC summary from rocprofv3 rocpd databases (one counter per pass).
Usage: rocpd_pmc.py fetch.db write.db [out.md]
HBM bytes per launch = 2 * FETCH_SIZE_KB * 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced
stream, MI355X_MICROARCH.md section HBM; checked here on standardize_cols_kernel, whose three passes over
the 8 GB fp64 X read 24 GB) + WRITE_SIZE_KB * 1024 (exact on the same kernel: 4 GB fp32 X written)."""
import sqlite3
import sys


def load(path, counter):
    db = sqlite3.connect(path)
    q = ("select kernel_name, count(*), avg(value), max(value) from counters_collection "
         "where counter_name = ? group by kernel_name")
    return {r[0]: r[1:] for r in db.execute(q, (counter,)).fetchall()}


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    lines = ["| kernel | launches | FETCH_SIZE max KB | WRITE_SIZE max KB | HBM read MB/launch (2x) | HBM write MB/launch |",
             "|---|---|---|---|---|---|"]
    for k in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
        if "admm::" not in k:
            continue
        fm = f[k][2]
        wm = w.get(k, (0, 0, 0))[2]
        lines.append(f"| `{k[:80]}` | {f[k][0]} | {fm:.0f} | {wm:.0f} | {2 * fm * 1024 / 1e6:.1f} | {wm * 1024 / 1e6:.1f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(__doc__ + "\n" + out + "\n")


if __name__ == "__main__":
    main()
