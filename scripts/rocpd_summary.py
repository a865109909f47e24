#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max,
and the idle gaps between consecutive dispatches.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        st = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    topn = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    for name, (c, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:topn]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {c} | {t/1e3:.3f} | {t/c:.2f} | {mn:.2f} | {mx:.2f} | {100*t/total:.1f} |")
    # gaps between consecutive dispatches (same process, ordered by start)
    gaps = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        g = (s1 - e0) / 1e3
        if g < 0 or g > 200:
            continue
        key = (n0.split("(")[0][-40:], n1.split("(")[0][-40:])
        st = gaps.setdefault(key, [0, 0.0])
        st[0] += 1
        st[1] += g
    lines.append("")
    lines.append("| gap after -> before | count | avg gap us |")
    lines.append("|---|---|---|")
    for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        lines.append(f"| `{a}` -> `{b}` | {c} | {t/c:.2f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
