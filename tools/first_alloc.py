#!/usr/bin/env python3
"""One fresh process, ONE state, no search in the caller: the NL / TL / AD kernel time the first allocation gets.
    python tools/first_alloc.py [KERNEL [NGPTOT]]      (CLOUDSC2_PLACE=0: without the allocator's placement)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "nl"
ngptot = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab), lregcl=(kind == "ad"))
nproma = int(os.environ.get("NPROMA", "128"))
nbk = (ngptot + nproma - 1) // nproma
one = os.environ.get("ONE_ARENA", "1") == "1" and kind != "nl"  # the perturbation set inside the state's own allocation
ds = c2.DeviceState.from_table(tab, nproma, ngptot, reserve=c2.FlatFields.pair_bytes(nbk, 137, nproma) if one else 0)
info = dict(ds.arena.info) if hasattr(ds.arena, "info") else {}
if kind == "nl":
    step = lambda: ds.nl(prm)  # noqa: E731
else:
    ds.satur(prm)
    inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device, arena=ds.arena if one else None)
    ds.increments(zero_supsat=(kind == "ad"), into=inc)
    if kind == "tl":
        step = lambda: ds.tl(prm, inc, dout)  # noqa: E731
    else:
        ds.tl(prm, inc, dout)
        step = lambda: ds.ad(prm, inc, dout, None)  # noqa: E731  (no cover-checkpoint plane without the evaporation branch)
for _ in range(20):
    step()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); step(); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print(json.dumps({"kernel": kind, "ngptot": ngptot, "place": os.environ.get("CLOUDSC2_PLACE", "1"), "ms_median": round(ms[15], 4),
                  "ms_min": round(ms[0], 4), "placement": info}))
