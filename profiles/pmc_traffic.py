#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-regex> <out.json>
Corrections (MI355X_MICROARCH.md, HBM section; re-calibrated in this repo on a float4 streaming copy): FETCH_SIZE counts
128-byte requests of 16 B/lane loads as 64 B on gfx950 -> x2; WRITE_SIZE is exact.  Both counters are in KiB."""
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter, rx):
    vals = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and rx.search(r["Kernel_Name"]):
            m = re.search(r"(\w+<[^>]*>|\w+)\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
            vals[(m.group(1) if m else r["Kernel_Name"][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return vals


def main():
    rx = re.compile(sys.argv[3])
    f = per_kernel(sys.argv[1], "FETCH_SIZE", rx)
    w = per_kernel(sys.argv[2], "WRITE_SIZE", rx)
    out = {"note": "per launch averages; FETCH_SIZE x2 (gfx950 16 B/lane correction), WRITE_SIZE exact; KiB -> bytes", "kernels": []}
    tot_b, tot_n = 0.0, 0
    for key in sorted(f):
        fv, wv = f[key], w.get(key, [])
        rd = 2.0 * 1024.0 * sum(fv) / len(fv)
        wr = 1024.0 * sum(wv) / len(wv) if wv else 0.0
        out["kernels"].append({"kernel": key[0], "grid": key[1], "launches": len(fv), "read_bytes": rd, "write_bytes": wr, "bytes": rd + wr})
        tot_b += (rd + wr) * len(fv)
        tot_n += len(fv)
    out["bytes_per_launch"] = tot_b / max(tot_n, 1)
    out["launches"] = tot_n
    for kv in sys.argv[5:]:
        k, v = kv.split("=", 1)
        out[k] = v
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
