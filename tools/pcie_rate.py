#!/usr/bin/env python3
"""PCIe-inclusive rate of the driver-level NL entry point (host arrays in, host arrays out): python tools/pcie_rate.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402

tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
for ngptot in (160000,):
    st = c2.state_from_table(tab, 128, ngptot)
    for rep in range(3):
        t0 = time.perf_counter()
        r = c2.run_state(prm, st, "nl")
        t1 = time.perf_counter()
        print(f"NL driver-level (host arrays) ngptot={ngptot} rep={rep}: wall {1e3 * (t1 - t0):.1f} ms -> "
              f"{ngptot / (t1 - t0):.3e} columns/s; kernel_ms={r}")
