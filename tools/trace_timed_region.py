#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED region of a bench.py run from a rocprofv3 --kernel-trace CSV.

bench.py first times a few launches on each candidate placement of the state (config.placement in its JSON line), then
runs W warm-up and K timed launches on the chosen one; `rocprofv3 --stats` averages over all of them.  This picks the
last K dispatches of the hot kernel -- the timed region -- and prints their statistics next to the all-launch ones.

    python tools/trace_timed_region.py KERNEL_TRACE.csv KERNEL_SUBSTR K
"""
import csv
import json
import statistics
import sys

path, pat, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
timed = dur[-k:]
out = {"kernel": rows[-1]["Kernel_Name"] if rows else None, "all_launches": {"calls": len(dur), "average_ns": statistics.mean(dur)},
       "timed_region": {"calls": len(timed), "average_ns": statistics.mean(timed), "min_ns": min(timed), "max_ns": max(timed),
                        "stddev_ns": statistics.pstdev(timed)}}
print(json.dumps(out, indent=1))
