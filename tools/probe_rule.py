import sys, ctypes as C
sys.path.insert(0, '/root/repo')
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
