#!/usr/bin/env python3
"""Does the placement effect belong to the AoSoA layout of B_LOC?  N states in different regions of the HBM; each is
timed with the real NL kernel (a) writing its tendencies into B_LOC as the reference lays it out (5 of 8 planes,
137 KiB every 1096 KiB), (b) writing them into five plane-major arrays (NPROMA,NLEV,NBLOCKS) allocated next to the
state, (c) additionally reading the four PGTEN* planes from plane-major arrays instead of B_CML.
    python tools/placement/placement_layout.py [N [NGPTOT [KERNEL]]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd.driver import _fld  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ngptot = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
gap = float(os.environ.get("GAP_GIB", "12"))
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
dev = torch.device("cuda:0")
states, spacers, soa = [], [], []
for i in range(n):
    used = torch.cuda.memory_reserved() / 2**30
    if i * gap > used + 1:
        spacers.append(torch.empty(int((i * gap - used) * 2**30), dtype=torch.uint8, device=dev))
    ds = c2.DeviceState.from_table(tab, 128, ngptot)
    states.append(ds)
    z = lambda: torch.zeros((ds.nb, ds.nlev, ds.nproma), dtype=torch.float64, device=dev)  # noqa: E731
    soa.append({"loc": [z() for _ in range(5)], "cml": [ds.B_CML[:, p].contiguous() for p in (0, 2, 3, 4)]})


def med(fn, warm=10, reps=9):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


def launch(ds, s, out_soa, in_soa):
    S = ds.nproma * ds.nlev
    i = ds.traj_inputs(False)
    o = ds.traj_outputs()
    zp = ds.zero_plane()
    if out_soa:
        o.tent, o.tenq, o.tenl, o.teni = (_fld(t, 0, S) for t in s["loc"][:4])
        zp = _fld(s["loc"][4], 0, S)
    if in_soa:
        i.gtent, i.gtenq, i.gtenl, i.gteni = (_fld(t, 0, S) for t in s["cml"])
    import ctypes as C
    B.check(B.lib.cloudsc2_nl_launch(C.byref(prm), ds.ptsphy, ds.nproma, ds.nlev, ds.ngptot, C.byref(i), C.byref(o), zp, 0.0,
                                     ds._stream(None)))


print("state  addr(B_LOC) GiB   AoSoA    SoA-out  SoA-out+in   (ms, NL kernel, %d columns)" % ngptot)
rows = []
for k, (ds, s) in enumerate(zip(states, soa)):
    t0 = med(lambda: launch(ds, s, False, False))
    t1 = med(lambda: launch(ds, s, True, False))
    t2 = med(lambda: launch(ds, s, True, True))
    rows.append((t0, t1, t2))
    print(f"{k:3d}    {ds.B_LOC.data_ptr() / 2**30:10.2f}    {t0:.4f}   {t1:.4f}   {t2:.4f}", flush=True)
import statistics as st
for j, nm in enumerate(("AoSoA", "SoA-out", "SoA-out+in")):
    v = [r[j] for r in rows]
    print(f"{nm:11s} min {min(v):.4f} median {st.median(v):.4f} max {max(v):.4f}")
