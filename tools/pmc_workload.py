#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes (HBM traffic of the NL / TL / AD kernels).

    CLOUDSC2_PLACE=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_workload.py
    CLOUDSC2_PLACE=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_workload.py

Each sweep of tools/pmc_plan.py is launched a few times on NGPTOT columns, in that order (CLOUDSC2_PLACE=0: the allocator
launches no sweeps of its own).  The SATUR kernel (reads 2 planes, writes 1 plane, 8 B per lane like every access of the
physics kernels) is the calibration dispatch with a known byte count, as /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) asks for access widths other than 16 B per lane.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from tools.pmc_plan import PLAN  # noqa: E402

assert os.environ.get("CLOUDSC2_PLACE") == "0", "run with CLOUDSC2_PLACE=0: the placement search launches NL sweeps of its own"
ngptot = int(os.environ.get("PMC_NGPTOT", "1048576"))  # state >> 256 MiB Infinity Cache
nproma = 128
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
ds = c2.DeviceState.from_table(tab, nproma, ngptot)  # tiled on the device: no 40 GB host copy at 1 M columns
inc = dout = None


def launch(label):
    global inc, dout
    if label == "satur":
        ds.satur(prm)
    elif label == "nl":
        ds.nl(prm)
    else:
        if inc is None:
            inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
            ds.increments(into=inc)
        if label == "tl":
            ds.tl(prm, inc, dout)  # stores the trajectory outputs: PFPLSL5 / PFPLSN5 for the reverse-only sweeps below
        else:
            ds.ad(prm, inc, dout, None, assign=label.endswith("assign"), sweep="reverse" if "reverse" in label else "both")


for label, reps in PLAN:
    for _ in range(reps):
        launch(label)
torch.cuda.synchronize()
print("pmc workload done", ngptot)
