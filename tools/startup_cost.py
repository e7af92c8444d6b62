#!/usr/bin/env python3
"""What a fresh process pays before its first sweep: imports, HIP initialisation, the state's allocation (placed by the library:
candidate allocations judged by the NL sweep itself; CLOUDSC2_PLACE=0 = plain hipMalloc) + tiling, the first and the second launch.
    python tools/startup_cost.py [NGPTOT]"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
t0 = time.time()
import torch
t1 = time.time()
import dwarf_p_cloudsc2_tl_ad_amd as c2
t2 = time.time()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t3 = time.time()
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
ds = c2.DeviceState.from_table(tab, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 160000)
torch.cuda.synchronize()
t4 = time.time()
ds.nl(prm); torch.cuda.synchronize()
t5 = time.time()
ds.nl(prm); torch.cuda.synchronize()
t6 = time.time()
print(json.dumps({"import_torch": t1 - t0, "import_pkg": t2 - t1, "cuda_init": t3 - t2, "state_alloc_place_expand": t4 - t3, "first_nl": t5 - t4, "second_nl": t6 - t5,
                  "place": os.environ.get("CLOUDSC2_PLACE", "1"), "info": dict(getattr(ds.arena, "info", {}))}))
