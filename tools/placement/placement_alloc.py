#!/usr/bin/env python3
"""The real kernels on states whose arrays come (a) from torch's allocator = plain hipMalloc, (b) from
cloudsc2_device_malloc = address ranges backed by hipMemCreate chunks.  N states of each kind, allocated alternately, each
timed; no placement search.    python tools/placement/placement_alloc.py [N [NGPTOT [KERNELS]]]"""
import os
import statistics as st
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ngptot = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
kernels = (sys.argv[3] if len(sys.argv) > 3 else "nl").split(",")
tab = c2.synthetic_table()


def med(fn, warm=10, reps=9):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


def workload(kind):
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=(kind == "ad"))
    ds = c2.DeviceState.from_table(tab, 128, ngptot)
    if kind == "nl":
        return (lambda: ds.nl(prm)), ds
    ds.satur(prm)
    inc = ds.increments(zero_supsat=(kind == "ad"))
    dout = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
    if kind == "tl":
        return (lambda: ds.tl(prm, inc, dout)), (ds, inc, dout)
    ds.tl(prm, inc, dout)
    scratch = ds.new_scratch()
    return (lambda: ds.ad(prm, inc, dout, scratch)), (ds, inc, dout, scratch)


for kind in kernels:
    res = {"torch": [], "library": []}
    keep = []
    for i in range(n):
        for how in ("torch", "library"):
            B.STATE_ALLOC_TORCH = how == "torch"
            w = workload(kind)
            keep.append(w)
            res[how].append(med(w[0]))
    for how, v in res.items():
        print(f"{kind} {ngptot} {how:8s} ms: " + " ".join(f"{x:.4f}" for x in v) + f"   median {st.median(v):.4f} min {min(v):.4f} max {max(v):.4f}", flush=True)
    del keep
    torch.cuda.empty_cache()
