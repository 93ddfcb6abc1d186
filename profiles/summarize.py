#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (rocprofv3 csv output) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pattern):
    hits = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def kernel_stats():
    f = find("trace", "*kernel_stats.csv")
    if not f:
        print("no kernel_stats.csv")
        return
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    with open(f) as fh:
        for i, row in enumerate(csv.DictReader(fh)):
            if i >= 12:
                break
            print(f"  {row.get('Name','?')[:60]:60s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} "
                  f"min_ns={row.get('MinNs')} max_ns={row.get('MaxNs')} pct={row.get('Percentage')}")


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print(f"no counter csv under {sub}")
        return
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(f"== counters ({sub}) per dispatch (mean over dispatches) ==")
    for k, d in acc.items():
        if "lfd_" not in k:
            continue
        for c, v in sorted(d.items()):
            print(f"  {k[:40]:40s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")


kernel_stats()
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mem"):
    counters(sub)
