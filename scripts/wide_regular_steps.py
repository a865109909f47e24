"""Durations of the wide solver's REGULAR x-update launches from a rocprofv3 kernel trace (csv): the launches of wide_x_kernel longer than
`floor_us` (an active-set launch takes 3 - 15 us, a regular step streams the matrix or its copy).  Usage: wide_regular_steps.py <dir> [floor_us]"""
import csv
import glob
import sys

d = sys.argv[1]
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wide_x_kernel" in r["Kernel_Name"]:
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
reg = sorted(x for x in rows if x > floor)
if reg:
    n = len(reg)
    print(f"{len(rows)} wide_x_kernel launches, {n} regular (> {floor} us): min {reg[0]:.1f} median {reg[n // 2]:.1f} mean {sum(reg) / n:.1f} p90 {reg[int(0.9 * n)]:.1f} max {reg[-1]:.1f} us")
else:
    print(f"{len(rows)} wide_x_kernel launches, none above {floor} us")
