#!/usr/bin/env python3
"""A/B of library builds with fresh processes (tools/first_alloc.py), interleaved, medians:
    python tools/ab_kernels.py "variants/a.so variants/b.so,ENV=1" "tl ad" "160000 1048576" [rounds]"""
import json
import os
import statistics
import subprocess
import sys

libs, kernels, sizes = sys.argv[1].split(), sys.argv[2].split(), sys.argv[3].split()
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
res = {}
for r in range(rounds):
    for k in kernels:
        for n in sizes:
            for lib in libs:
                parts = lib.split(",")
                env = dict(os.environ, CLOUDSC2_LIB=os.path.abspath(parts[0]), **dict(kv.split("=") for kv in parts[1:]))
                out = subprocess.run([sys.executable, "tools/first_alloc.py", k, n], env=env, capture_output=True, text=True, timeout=300)
                if out.returncode != 0:
                    print("FAILED", lib, k, n, out.stderr[-500:], flush=True)
                    continue
                d = json.loads(out.stdout.strip().split("\n")[-1])
                res.setdefault((k, n, lib), []).append(d["ms_median"])
for (k, n, lib), v in sorted(res.items()):
    bpc = {"nl": 28536, "tl": 57072, "ad": 85608}[k]
    med = statistics.median(v)
    print(f"{k} {n:>8} {lib:44s} median {med:7.4f} ms  frac {bpc * int(n) / (med * 1e-3) / 8e12:5.3f}   all {[round(x, 4) for x in v]}", flush=True)
