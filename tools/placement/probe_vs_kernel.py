#!/usr/bin/env python3
"""Do the allocator's probes predict the NL kernel's time on a state built in the same memory?  ONE process, K unplaced arenas kept
alive (CLOUDSC2_PLACE=0 -> plain hipMalloc, consecutive places in the HBM); on each: the write probe, the mixed (NL-pattern) probe,
then a state is built in it and the real NL kernel is timed.  usage: CLOUDSC2_PLACE=0 python tools/probe_vs_kernel.py [K] [NGPTOT]"""
import os
import sys

os.environ["CLOUDSC2_PLACE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ngptot = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
dev = torch.device("cuda:0")
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
stream = torch.cuda.current_stream(dev)
probes = []
orig_init = B.DeviceArena.__init__


def probing_init(self, nbytes, device="cuda:0"):
    orig_init(self, nbytes, device)
    if self.buf is not None and nbytes > (1 << 28):
        probes.append((B.device_probe(self.buf.ptr, nbytes, 0, 5), B.device_probe(self.buf.ptr, nbytes, 1, 5)))


B.DeviceArena.__init__ = probing_init
keep, rows = [], []
for k in range(K):
    ds = c2.DeviceState.from_table(tab, 128, ngptot, dev)
    keep.append(ds)
    for _ in range(20):
        ds.nl(prm, stream)
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    ev[0].record(stream)
    for i in range(30):
        ds.nl(prm, stream)
        ev[i + 1].record(stream)
    torch.cuda.synchronize(dev)
    ms = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(30)]))
    w, m = probes[-1]
    rows.append((w, m, ms))
    print(f"arena {k:2d}: write probe {w:.4f} ms  mixed probe {m:.4f} ms  NL kernel {ms:.4f} ms", flush=True)
a = np.array(rows)
print("correlation with the NL kernel's time: write probe %.3f, mixed probe %.3f" % (np.corrcoef(a[:, 0], a[:, 2])[0, 1], np.corrcoef(a[:, 1], a[:, 2])[0, 1]))
print("NL kernel ms: min %.4f max %.4f; the arena the write probe would pick: %.4f, the mixed probe: %.4f" %
      (a[:, 2].min(), a[:, 2].max(), a[a[:, 0].argmin(), 2], a[a[:, 1].argmin(), 2]))
