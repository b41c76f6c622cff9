#!/usr/bin/env python3
"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel summary committed under
profiles/.   usage: rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-92s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%")]
    for name, n, s, a, mn, mx in rows:
        lines.append("%-92s %8d %12.3f %10.1f %10.1f %10.1f %6.2f" % (name[:92], n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    lines.append("total kernel time %.3f ms" % (tot / 1e6))
    # counters, when the run collected any (--pmc)
    try:
        pmc = list(cur.execute("select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
                               "group by k.name, p.counter_name order by k.name"))
        if pmc:
            lines.append("")
            lines.append("%-80s %-28s %8s %16s %14s" % ("kernel", "counter", "n", "sum", "avg"))
            for name, c, n, s, a in pmc:
                lines.append("%-80s %-28s %8d %16.0f %14.1f" % (name[:80], c, n, s, a))
    except sqlite3.Error as e:
        lines.append("(no counter tables: %s)" % e)
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
