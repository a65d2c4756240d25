#!/usr/bin/env python3
"""Mean per launch of every counter of a kernel in rocprofv3 counter_collection csv files: pmc_sum.py <kernel substring> <csv>..."""
import csv, sys, collections
pat = sys.argv[1]
for f in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in acc.items():
        print(f"   {k:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
