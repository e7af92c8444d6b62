#!/usr/bin/env python3
"""One CSV with a row per timed kernel from rocprofv3 --kernel-trace runs of bench.py (tools/evidence.sh).

    python tools/kernel_stats_rows.py OUT.csv  LABEL:TRACE.csv:KERNEL_SUBSTR:K:NGPTOT:BYTES_PER_COLUMN ...

For every run the last K dispatches of the kernel whose name contains KERNEL_SUBSTR are the timed region (what comes before is the
placement search's probes, the warm-up and the settle launches); the row carries their count, average / min / max duration and the
roofline fraction that average implies -- BYTES_PER_COLUMN x NGPTOT / average / 8 TB/s -- so that every figure of DESIGN.md's
numbers table can be recomputed from profiles/ alone.  BYTES_PER_COLUMN 0: no fraction (the Taylor sweep is a whole test, not a
streaming kernel with an algorithmic byte count)."""
import csv
import statistics
import sys

out = sys.argv[1]
rows = []
for spec in sys.argv[2:]:
    label, path, pat, k, ngptot, bpc = spec.split(":")
    k, ngptot, bpc = int(k), int(ngptot), int(bpc)
    try:
        tr = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
    except OSError as e:
        print("skip", label, e, file=sys.stderr)
        continue
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
    if not dur:
        print("skip", label, "no dispatch of", pat, file=sys.stderr)
        continue
    timed = dur[-k:]
    avg = statistics.mean(timed)
    rows.append({"label": label, "kernel": tr[-1]["Kernel_Name"][:100], "ngptot": ngptot, "dispatches_all": len(dur),
                 "timed_calls": len(timed), "timed_avg_ns": round(avg, 1), "timed_min_ns": min(timed), "timed_max_ns": max(timed),
                 "all_avg_ns": round(statistics.mean(dur), 1), "bytes_per_column": bpc,
                 "frac_of_8TBps": round(bpc * ngptot / (avg * 1e-9) / 8e12, 4) if bpc else ""})
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
for r in rows:
    print(r["label"], r["ngptot"], r["timed_calls"], "x", r["timed_avg_ns"] / 1e6, "ms", r["frac_of_8TBps"])
