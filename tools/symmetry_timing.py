#!/usr/bin/env python3
"""Device time of the adjoint test (cloudsc2_state_ad_symmetry) on a resident state: as the library runs it (`fused`: SATUR, the TL
sweep that forms its increments and <y,y>, the reverse sweep alone in its assign form that forms <x0,x_adj> and norm3) and as the
plain sequence (`unfused`, CLOUDSC2_AD_SYMMETRY_FULL=1, measurements only: TL, norm-1 kernel, CLOUDSC2AD's two sweeps, norm-2/3
kernel).     python tools/symmetry_timing.py NGPTOT [NPROMA]"""
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
for label, env in (("fused", "0"), ("unfused", "1"), ("fused_again", "0")):
    os.environ["CLOUDSC2_AD_SYMMETRY_FULL"] = env
    ms, zn = [], None
    for _ in range(8):
        zn, ok, t = rs.ad_symmetry(prm)
        assert ok, zn
        ms.append(t)
    out[label] = {"kernel_ms_median": float(np.median(ms[2:])), "znormg_eps": zn}
out["whole_test_speedup"] = out["unfused"]["kernel_ms_median"] / out["fused"]["kernel_ms_median"]
print(json.dumps(out))
