#!/usr/bin/env python3
"""Is a slow placement the sum of many small effects or the fault of single arrays?  Take the fastest (F) and the slowest
(S) of N states, then time hybrids: arrays 0..k from F, the rest from S, and single-array swaps both ways."""
import copy
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
names = list(c2.DeviceState.FULL + c2.DeviceState.HALF) + ["B_CML", "B_LOC", "PCLV"]


def timeit(ds, warm=10, reps=7):
    for _ in range(warm):
        ds.nl(prm)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); ds.nl(prm); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]


t = [timeit(s) for s in states]
print("states:", " ".join(f"{x:.3f}" for x in t))
F, S = states[min(range(n), key=lambda i: t[i])], states[max(range(n), key=lambda i: t[i])]
print(f"F {min(t):.4f}  S {max(t):.4f}")
h = copy.copy(S)
print("prefix hybrids (arrays 0..k from F):")
for k, nm in enumerate(names):
    setattr(h, nm, getattr(F, nm))
    print(f"  +{nm:9s} {timeit(h, 5, 5):.4f}")
print("single swaps: S with one array from F | F with one array from S")
for nm in names:
    a, b = copy.copy(S), copy.copy(F)
    setattr(a, nm, getattr(F, nm))
    setattr(b, nm, getattr(S, nm))
    print(f"  {nm:9s} {timeit(a, 5, 5):.4f} | {timeit(b, 5, 5):.4f}")
