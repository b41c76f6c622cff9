#!/usr/bin/env python3
"""rocprofv3 rocpd kernel trace -> per (kernel, grid) statistics: tells the pyramid octaves apart.
usage: rocpd_by_grid.py <results.db> [min_total_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, grid_x, grid_y, grid_z, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by 1,2,3,4 order by 6 desc"))
tot = sum(r[5] for r in rows) or 1
print("%-70s %-18s %7s %10s %9s %9s %6s" % ("kernel", "grid", "calls", "total_ms", "avg_us", "min_us", "%"))
for name, gx, gy, gz, n, s, a, mn in rows:
    if s / 1e6 < thr:
        continue
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:70]
    print("%-70s %-18s %7d %10.2f %9.1f %9.1f %6.2f" % (short, "%dx%dx%d" % (gx, gy, gz), n, s / 1e6, a / 1e3, mn / 1e3, 100.0 * s / tot))
print("total %.2f ms" % (tot / 1e6))
