"""usage: python tools/probe_rule.py  -- on a GPU box: cloudsc2_dispatch_probe six times before and three times after an NL launch
(waves checked / waves off the predicted SIMD population)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dwarf_p_cloudsc2_tl_ad_amd as c2
from dwarf_p_cloudsc2_tl_ad_amd import binding as B
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
for k in range(6):
    a, b = C.c_longlong(), C.c_longlong()
    B.check(B.lib.cloudsc2_dispatch_probe(C.byref(a), C.byref(b)))
    print('probe', k, a.value, b.value, flush=True)
tab = c2.synthetic_table(); prm = c2.default_params(c2.ceta_from_table(tab))
ds = c2.DeviceState.from_table(tab, 128, 160000)
ds.nl(prm); torch.cuda.synchronize()
for k in range(3):
    a, b = C.c_longlong(), C.c_longlong()
    B.check(B.lib.cloudsc2_dispatch_probe(C.byref(a), C.byref(b)))
    print('probe after NL', k, a.value, b.value, flush=True)
