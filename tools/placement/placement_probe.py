#!/usr/bin/env python3
"""Does the placement of the state in HBM change the NL kernel's time?  Several states are allocated side by side in
one process (so they occupy different physical memory) and each is timed: python tools/placement/placement_probe.py [NGPTOT [N]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
ncopies = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
states = [c2.DeviceState.from_table(tab, 128, ngptot) for _ in range(ncopies)]
torch.cuda.synchronize()
for rnd in range(2):
    line = []
    for ds in states:
        for _ in range(3):
            ds.nl(prm)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for a, b in ev:
            a.record(); ds.nl(prm); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        line.append(ms[7])
    print(f"round {rnd}: " + " ".join(f"{m:.3f}" for m in line), flush=True)
