#!/usr/bin/env python3
"""How busy is the chip?  From a rocprofv3 rocpd kernel trace: wall span, union of intervals with >=1 kernel running,
union of the chip-filling kernels (blur*, extrema), and the summed durations per class.
usage: rocpd_overlap.py <results.db> [t0_frac t1_frac]   (optional window of the trace, as fractions of its span)"""
import sqlite3
import sys


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    if len(sys.argv) > 3:
        a, b = float(sys.argv[2]), float(sys.argv[3])
        t0, t1 = t0 + a * (t1 - t0), t0 + b * (t1 - t0)
    rows = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    span = t1 - t0
    heavy = lambda n: ("blur" in n) or ("extrema" in n)         # blur_stream (r01-r02), blur16_stream / blur16_tile (r03+), extrema_stream / extrema_kernel
    print("window %.1f ms, %d kernels" % (span / 1e6, len(rows)))
    print("any kernel running   : %5.1f %% of the window" % (100.0 * union([(s, e) for _, s, e in rows]) / span))
    print("blur/extrema running : %5.1f %% of the window" % (100.0 * union([(s, e) for n, s, e in rows if heavy(n)]) / span))
    print("sum of blur/extrema durations / window: %.2f   sum of all durations / window: %.2f" % (
        sum(e - s for n, s, e in rows if heavy(n)) / span, sum(e - s for _, s, e in rows) / span))


if __name__ == "__main__":
    main()
