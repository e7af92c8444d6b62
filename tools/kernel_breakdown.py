#!/usr/bin/env python3
"""Per-kernel time of a rocprofv3 --kernel-trace run from its database (rocprofv3 -d DIR -o NAME writes DIR/NAME_results.db):
    python tools/kernel_breakdown.py DIR/NAME_results.db [GRID_THREADS ...]      (only dispatches of these grid sizes, if given)"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
grids = {int(a) for a in sys.argv[2:]}
agg = collections.OrderedDict()
for name, s, e, g in c.execute("select name, start, end, grid_x from kernels order by start"):
    if grids and g not in grids:
        continue
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", name)
    agg.setdefault((m.group(1) if m else name[:40], g), []).append((e - s) / 1e6)
for (n, g), v in agg.items():
    print(f"{n:30s} grid {g:>10} threads  n={len(v):4d}  avg {sum(v) / len(v):8.4f} ms  min {min(v):8.4f}  total {sum(v):9.3f}")
