#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean of each counter."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "das_fused" in k or "bf::" in k:
                acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{k:60s} {c:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
