#!/usr/bin/env python3
"""Sum of one PMC counter per kernel and grid size from a rocprofv3 --pmc run written with --output-format csv:
    python tools/pmc_kernel.py DIR KERNEL_SUBSTRING     (FETCH_SIZE: units of the guide's HBM section -- 64 B on gfx950 after its correction)"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg.setdefault((r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size"]), r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, g, c), v in agg.items():
    print(f"{k:40s} grid {g:>10} {c:12s} dispatches {len(v):3d}  mean {sum(v) / len(v):.6g}")
