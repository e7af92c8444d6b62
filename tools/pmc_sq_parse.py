#!/usr/bin/env python3
"""Average per-dispatch SQ counters of one kernel from rocprofv3 --pmc csv output dirs: pmc_sq_parse.py KERNEL_SUBSTR dir..."""
import csv
import glob
import os
import sys

pat = sys.argv[1]
acc = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} {sum(v.values()) / len(v):18.1f}  (n={len(v)})")
