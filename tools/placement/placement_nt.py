#!/usr/bin/env python3
"""Fast and slow placements under other builds of the library (e.g. without non-temporal stores), same allocations:
python tools/placement/placement_nt.py N lib1.so lib2.so ..."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

n = int(sys.argv[1])
libs = [("default", B.lib)]
for path in sys.argv[2:]:
    l = C.CDLL(os.path.abspath(path))
    l.cloudsc2_nl_launch.argtypes = B.lib.cloudsc2_nl_launch.argtypes
    l.cloudsc2_nl_launch.restype = C.c_int
    libs.append((os.path.basename(path), l))
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
states, spacers = [], []
for i in range(n):
    used = torch.cuda.memory_reserved() / 2**30
    if i * 12 > used + 1:
        spacers.append(torch.empty(int((i * 12 - used) * 2**30), dtype=torch.uint8, device="cuda"))
    states.append(c2.DeviceState.from_table(tab, 128, 160000))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def med(lib, ds, warm=10, reps=7):
    i, o, zp = ds.traj_inputs(False), ds.traj_outputs(), ds.zero_plane()
    run = lambda: lib.cloudsc2_nl_launch(C.byref(prm), ds.ptsphy, ds.nproma, ds.nlev, ds.ngptot, C.byref(i), C.byref(o), zp, 0.0, stream)  # noqa: E731
    for _ in range(warm):
        assert run() == 0
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


for name, lib in libs:
    print(f"{name:14s}", " ".join(f"{med(lib, ds):.3f}" for ds in states))
