#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel:  python scripts/pmc_summary.py <dir> [name-filter]
Reads *_counter_collection.csv or *_results.db (rocprofv3's default sqlite output) below <dir>."""
import csv, glob, os, sqlite3, sys
from collections import defaultdict


def rows(root):
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            yield f, r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"], float(r["Counter_Value"]), 0
    for f in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for r in con.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
            yield f, r[0], r[1], r[2], float(r[3]), r[4]


def main():
    root, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc, cnt, dur = defaultdict(float), defaultdict(set), defaultdict(list)
    for f, k, d, c, v, t in rows(root):
        if flt and flt not in k:
            continue
        k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
        acc[(k, c)] += v
        if (f, d) not in cnt[(k, c)]:
            cnt[(k, c)].add((f, d))
            dur[(k, c)].append(t)
    for (k, c) in sorted(acc):
        n = len(cnt[(k, c)])
        print(f"{k:60s} {c:28s} n={n:3d} mean={acc[(k, c)] / n:16.1f}  dur_us={sum(dur[(k, c)]) / n / 1e3:10.1f}")


if __name__ == "__main__":
    main()
