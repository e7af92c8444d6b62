#!/usr/bin/env python3
"""Per state: NL time next to plain fills / reads of its two big buffers -- is a slow B_LOC slow for any writer?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
states, spacers = [], []
for i in range(n):
    used = torch.cuda.memory_reserved() / 2**30
    if i * 12 > used + 1:
        spacers.append(torch.empty(int((i * 12 - used) * 2**30), dtype=torch.uint8, device="cuda"))
    states.append(c2.DeviceState.from_table(tab, 128, 160000))


def med(fn, warm=5, reps=7):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


print("state  NL     fill(B_LOC) GB/s   fill(B_CML) GB/s   sum(B_LOC) GB/s  sum(B_CML) GB/s   fill(5 planes of B_LOC)")
for k, ds in enumerate(states):
    gb = ds.B_LOC.numel() * 8 / 1e9
    t_nl = med(lambda: ds.nl(prm), 10)
    t_fl = med(lambda: ds.B_LOC.zero_())
    t_fc = med(lambda: ds.B_CML.fill_(1e-9))
    t_sl = med(lambda: ds.B_LOC.sum())
    t_sc = med(lambda: ds.B_CML.sum())
    t_p = med(lambda: [ds.B_LOC[:, p].zero_() for p in (0, 2, 3, 4, 7)])
    print(f"{k:3d}  {t_nl:.3f}   {t_fl:.3f} {gb / t_fl * 1e3:6.0f}    {t_fc:.3f} {gb / t_fc * 1e3:6.0f}    {t_sl:.3f} {gb / t_sl * 1e3:6.0f}   {t_sc:.3f} {gb / t_sc * 1e3:6.0f}    {t_p:.3f}")
