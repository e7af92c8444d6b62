#!/usr/bin/env python3
"""Per-launch device time of the first 40 NL launches on a fresh state, then of 10 more after a 2 s pause: the slow first launches are
the GPU coming out of idle, not the allocation (0.93 0.84 0.84 0.86 0.89 0.87 ... 0.82 ms, and 0.96 0.86 0.88 0.90 ... again after the
pause) -- why bench.py and the resident Fortran main launch 15 times before they time.  usage: python tools/settle_series.py [NGPTOT]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
dev = torch.device("cuda:0")
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
stream = torch.cuda.current_stream(dev)
ds = c2.DeviceState.from_table(tab, 128, ngptot, dev)
torch.cuda.synchronize(dev)


def series(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(stream)
    for i in range(n):
        ds.nl(prm, stream)
        ev[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


print("first 40 launches (ms):", " ".join(f"{x:.3f}" for x in series(40)))
time.sleep(2.0)
print("after a 2 s pause     :", " ".join(f"{x:.3f}" for x in series(10)))
