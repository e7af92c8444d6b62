#!/usr/bin/env python3
"""A/B comparison of library builds that is robust against box-to-box and run-to-run drift: the variants are run
interleaved for several rounds and the median per variant is reported.
    python tools/ab_bench.py "libA.so libB.so" "nl tl" "160000 1048576" [rounds [extra bench.py arguments, e.g. --levapls2]]"""
import json
import os
import statistics
import subprocess
import sys

# a "lib" may carry environment settings: "path.so,NAME=VALUE,..."
libs, kernels, sizes = sys.argv[1].split(), sys.argv[2].split(), sys.argv[3].split()
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
extra = sys.argv[5].split() if len(sys.argv) > 5 else []
res = {}
for r in range(rounds):
    for k in kernels:
        for n in sizes:
            for lib in libs:
                parts = lib.split(",")
                env = dict(os.environ, CLOUDSC2_LIB=parts[0], **dict(kv.split("=") for kv in parts[1:]))
                out = subprocess.run([sys.executable, "bench.py", "--kernel", k, "--ngptot", n, "--steps", "40" if int(n) > 500000 else "200",
                                      "--warmup", "5", "--no-cpu-baseline", "--no-companions"] + extra, env=env, capture_output=True,
                                     text=True, timeout=300).stdout
                d = json.loads(out.strip().split("\n")[-1])
                res.setdefault((k, n, lib), []).append(d["roofline"]["kernel_ms_avg"])
for (k, n, lib), v in sorted(res.items()):
    bpc = {"nl": 28536, "tl": 57072, "ad": 85608}[k]
    med = statistics.median(v)
    print(f"{k} {n:>8} {lib:40s} median {med:7.3f} ms  frac {bpc * int(n) / (med * 1e-3) / 8e12:5.3f}   all {[round(x, 3) for x in v]}")
