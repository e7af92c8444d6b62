"""usage: python tools/debug/first_call.py [NGPTOT]  -- where the first host-array driver call's time goes: workspace allocation, first touch of the
caller's arrays by the DMA engine, everything else."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
tab = c2.synthetic_table(); prm = c2.default_params(c2.ceta_from_table(tab))
t = time.perf_counter; ms = lambda a: round((t() - a) * 1e3, 1)  # noqa: E731
a = t(); B.check(B.lib.cloudsc2_device_prepare()); print("device_prepare", ms(a), "ms")
p = C.c_void_p()
os.environ["CLOUDSC2_PLACE"] = "0"
for k in range(2):
    a = t(); B.check(B.lib.cloudsc2_device_malloc(C.byref(p), C.c_size_t(6494720000))); print("plain hipMalloc of 6.5 GB", ms(a), "ms")
    a = t(); B.check(B.lib.cloudsc2_device_free(p)); print("  free", ms(a), "ms")
st = c2.state_from_table(tab, 128, n)
for k in range(3):
    a = t(); km = c2.run_state(prm, st, "nl"); print(f"cloudsc2_nl_run call {k}", ms(a), "ms  (kernels", round(float(km), 2), "ms)")
B.lib.cloudsc2_release_workspace()
a = t(); c2.run_state(prm, st, "nl"); print("after cloudsc2_release_workspace (workspace allocated again, arrays already seen)", ms(a), "ms")
st2 = c2.state_from_table(tab, 128, n)
a = t(); c2.run_state(prm, st2, "nl"); print("NEW host arrays (workspace kept)", ms(a), "ms")
a = t(); c2.run_state(prm, st2, "nl"); print("  again", ms(a), "ms")
