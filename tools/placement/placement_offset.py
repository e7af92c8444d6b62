#!/usr/bin/env python3
"""Is the slow/fast property of a B_LOC placement a matter of address bits?  One 4.5 GiB allocation; B_LOC views at
different offsets inside it; strided plane fills (the NL write pattern) and the NL kernel timed for each."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
ds = c2.DeviceState.from_table(tab, 128, 160000)
nb, nlev, nproma = ds.nb, ds.nlev, ds.nproma
S = nproma * nlev
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def med(fn, warm=5, reps=7):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


def nl_with(buf):
    i = ds.traj_inputs(False)
    o = ds.traj_outputs()
    f = lambda p: B.Field(buf.data_ptr() + 8 * p * S, 8 * S)  # noqa: E731
    o.tent, o.tenq, o.tenl, o.teni = f(0), f(2), f(3), f(4)
    zp = f(7)
    return lambda: B.check(B.lib.cloudsc2_nl_launch(C.byref(prm), ds.ptsphy, nproma, nlev, ds.ngptot, C.byref(i), C.byref(o), zp, 0.0, stream))


MiB = 1 << 20
for trial in range(3):
    big = torch.zeros(int(4.5 * 2**30) // 8, dtype=torch.float64, device="cuda")
    print(f"arena {trial}: base {big.data_ptr():#x}")
    for off in (0, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 3072, 6, 1030):
        v = big[off * MiB // 8: off * MiB // 8 + nb * 8 * S].view(nb, 8, nlev, nproma)
        t_p = med(lambda: [v[:, p].zero_() for p in (0, 2, 3, 4, 7)])
        t_nl = med(nl_with(v), 10)
        print(f"   offset {off:5d} MiB: plane fills {t_p:.3f}  NL {t_nl:.3f}")
    spacer = torch.empty(int(20 * 2**30), dtype=torch.uint8, device="cuda")  # next arena elsewhere
