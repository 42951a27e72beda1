#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter).  usage: pmc_sum.py <dir> [kernel substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel"
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        acc[(k[:48], r["Counter_Name"])] += float(r["Counter_Value"]); n[(k[:48], r["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:50s} {c:36s} {v:16.6g}  dispatches {n[(k,c)]}")
