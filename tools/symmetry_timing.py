#!/usr/bin/env python3
"""Device time of the adjoint test (cloudsc2_state_ad_symmetry: SATUR, increments, TL, norm 1, AD, norms 2-3) on a resident
state, with the AD leg as the reverse sweep alone (what the library does) and as CLOUDSC2AD's two sweeps
(CLOUDSC2_AD_SYMMETRY_FULL=1, measurements only).     python tools/symmetry_timing.py NGPTOT [NPROMA]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

ngptot = int(sys.argv[1])
nproma = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
rs = c2.ResidentState.from_table(tab, nproma, ngptot)
out = {"ngptot": ngptot, "nproma": nproma}
for label, env in (("reverse_only", "0"), ("both_sweeps", "1"), ("reverse_only_again", "0")):
    os.environ["CLOUDSC2_AD_SYMMETRY_FULL"] = env
    ms, zn = [], None
    for _ in range(8):
        zn, ok, t = rs.ad_symmetry(prm)
        assert ok, zn
        ms.append(t)
    out[label] = {"kernel_ms_median": float(np.median(ms[2:])), "znormg_eps": zn}
out["whole_test_speedup"] = out["both_sweeps"]["kernel_ms_median"] / out["reverse_only"]["kernel_ms_median"]
print(json.dumps(out))
