#!/usr/bin/env python3
"""One process, one placed state (default 1 048 576 columns): the NL kernel timed again and again -- with the bench's parameter set
(CETA from the table) and with the parameter set the allocator's kernel probe uses (default constants, linear CETA, its constants
table was created while the candidate pool was still allocated) -- to see whether a time that differs from the allocator's own
measurement is there from the start, drifts, or depends on the parameter set.  usage: python tools/one_process_series.py [NGPTOT] [ROUNDS]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
tab = c2.synthetic_table()
stream = torch.cuda.current_stream(dev)
prm_bench = c2.default_params(c2.ceta_from_table(tab))
nlev_t = tab["PT"].shape[0]
prm_probe = c2.default_params([(k + 0.5) / nlev_t for k in range(nlev_t)])
prm_probe.math_mode = 1
ds = None


def time_nl(prm, n=20, settle=5):
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


def trace(state, stage):
    global ds
    ds = state
    print(f"  state build, {stage}: NL {time_nl(prm_probe, n=10, settle=2):.4f} ms", flush=True)


c2.DeviceState._trace = staticmethod(trace)
ds = c2.DeviceState.from_table(tab, 128, ngptot, dev)
c2.DeviceState._trace = None
print("allocator:", ds.arena.info, flush=True)
for r in range(rounds):
    a = time_nl(prm_bench)
    b = time_nl(prm_probe)
    print(f"round {r}: NL with the bench's parameters {a:.4f} ms | with the probe's parameters {b:.4f} ms", flush=True)
    time.sleep(0.3)
