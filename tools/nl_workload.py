#!/usr/bin/env python3
"""NL-only workload for rocprofv3 counter passes: python3 tools/nl_workload.py [ngptot] [kernel nl|tl|ad] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
kernel = sys.argv[2] if len(sys.argv) > 2 else "nl"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
ds = c2.DeviceState.from_table(tab, int(os.environ.get("NPROMA", "128")), ngptot)
if kernel == "nl":
    for _ in range(reps):
        ds.nl(prm)
else:
    ds.satur(prm)
    inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    ds.increments(into=inc)
    if kernel == "tl":
        for _ in range(reps):
            ds.tl(prm, inc, dout)
    else:
        ds.tl(prm, inc, dout)
        scratch = ds.new_scratch()
        for _ in range(reps):
            ds.ad(prm, inc, dout, scratch)
torch.cuda.synchronize()
