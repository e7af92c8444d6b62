#!/usr/bin/env python3
"""Is the slow class of a place a property of the place alone, or of the place AND the data?  ONE process, K unplaced states kept alive
(CLOUDSC2_PLACE=0: consecutive plain hipMallocs).  On each: the NL kernel on the real state; then the whole arena is cleared and the same
kernel runs on all-zero fields (same loads and stores, same memory); then the real state is restored and timed again; then the arena is
filled with random bits (as float64 NaN-free patterns) and timed once more.  usage: python tools/data_dependence.py [K] [NGPTOT]"""
import os
import sys

if os.environ.get("DD_PLACE") != "1":  # DD_PLACE=1: through the placing allocator (one state, e.g. at 1 048 576 columns)
    os.environ["CLOUDSC2_PLACE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ngptot = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
dev = torch.device("cuda:0")
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
stream = torch.cuda.current_stream(dev)


def time_nl(ds, n=30, settle=20):
    for _ in range(settle):
        ds.nl(prm, stream)
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(stream)
    for i in range(n):
        ds.nl(prm, stream)
        ev[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))


keep = []
for k in range(K):
    ds = c2.DeviceState.from_table(tab, 128, ngptot, dev)
    keep.append(ds)
    raw = ds.arena.raw
    t_real = time_nl(ds)
    saved = raw.clone()
    raw.zero_()
    t_zero = time_nl(ds)
    raw.copy_(saved)
    t_real2 = time_nl(ds)
    # random mantissas on a fixed exponent: finite doubles in [1, 2) everywhere (inputs and outputs alike)
    v = raw.view(torch.int64)
    v.random_(0, 1 << 52)
    v.bitwise_or_(0x3FF0000000000000)
    t_rand = time_nl(ds)
    raw.copy_(saved)
    del saved, v
    print(f"state {k:2d}: NL on the real state {t_real:.4f} ms | on all-zero fields {t_zero:.4f} ms | real state restored {t_real2:.4f} ms | "
          f"on random doubles in [1,2) {t_rand:.4f} ms", flush=True)
