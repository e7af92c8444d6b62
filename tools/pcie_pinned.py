#!/usr/bin/env python3
"""Does page-locking the caller's arrays ONCE (hipHostRegister) change the PCIe-inclusive rate of the host-array NL driver?
    python tools/pcie_pinned.py [NGPTOT]
Prints the wall time of cloudsc2_nl_run per call before and after registering the 18 host arrays, and what the registration cost."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
st = c2.state_from_table(tab, 128, ngptot)


def calls(label, n=4):
    for rep in range(n):
        t0 = time.perf_counter()
        k = c2.run_state(prm, st, "nl")
        dt = time.perf_counter() - t0
        print(f"{label} rep={rep}: wall {1e3 * dt:.1f} ms -> {ngptot / dt:.3e} columns/s; kernel_ms={k:.3f}", flush=True)


calls("pageable")
rt = torch.cuda.cudart()
t0 = time.perf_counter()
nbytes = 0
for a in st.driver_arrays():
    rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    assert int(rc) == 0, rc
    nbytes += a.nbytes
print(f"hipHostRegister of {nbytes / 1e9:.2f} GB in 18 arrays: {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
calls("registered")
for a in st.driver_arrays():
    rt.cudaHostUnregister(a.ctypes.data)
calls("pageable again", 2)
