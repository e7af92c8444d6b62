#!/usr/bin/env python3
"""Kernel time of the whole Taylor test (cloudsc2_state_tl_taylor on a resident state) for one library build:
    CLOUDSC2_LIB=path.so python tools/taylor_ab.py [MATH_MODE]      (1 fast, 2 precise = the driver's default)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tab = c2.synthetic_table()
res = {}
for nproma, ngptot in ((1, 100), (32, 100), (128, 16384), (128, 160000), (32, 160000), (128, 1048576)):
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    prm.math_mode = mode
    rs = c2.ResidentState.from_table(tab, nproma, ngptot)
    ms = []
    for _ in range(4):
        z, ok, itest, t = rs.tl_taylor(prm)
        ms.append(t)
    res[f"{ngptot}x{nproma}"] = {"ms": round(min(ms[1:]), 3), "passed": bool(ok), "penalty": int(itest), "ratio6": float(z[5])}
    del rs
print(json.dumps({"lib": os.environ.get("CLOUDSC2_LIB", "default"), "math_mode": mode, "taylor_test": res}))
