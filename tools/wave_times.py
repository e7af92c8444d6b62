#!/usr/bin/env python3
"""When do the waves of one NL launch start and finish, and where do they run?  Needs a diagnostic build of the library:
    hipcc <CXXFLAGS of csrc/Makefile> -DC2_WAVE_TIMES -DC2_SINGLE_TU -shared -o /tmp/wt.so csrc/cloudsc2_kernels.hip
    CLOUDSC2_LIB=/tmp/wt.so python tools/wave_times.py [NGPTOT [NPROMA [nl|tl|ad|ad_reverse]]]
Prints the distribution of the waves' start and end times (microseconds after the first wave's start; 100 MHz clock, 10 ns
resolution) for a launch in steady state, per XCD and per number of waves sharing a SIMD."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402

ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
nproma = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tab = c2.synthetic_table()
prm = c2.default_params(c2.ceta_from_table(tab))
kind = sys.argv[3] if len(sys.argv) > 3 else "nl"
ds = c2.DeviceState.from_table(tab, nproma, ngptot)
nwaves = (ds.nb * nproma + 127) // 128 * 2
if kind == "nl":
    step = lambda: ds.nl(prm)  # noqa: E731
else:
    ds.satur(prm)
    inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    ds.increments(zero_supsat=(kind != "tl"), into=inc)
    ds.tl(prm, inc, dout)
    step = {"tl": lambda: ds.tl(prm, inc, dout), "ad": lambda: ds.ad(prm, inc, dout, None),
            "ad_reverse": lambda: ds.ad(prm, inc, dout, None, sweep="reverse")}[kind]
for _ in range(30):
    step()
torch.cuda.synchronize()
log = B.lib.cloudsc2_debug_wave_log
log.argtypes = [C.c_void_p, C.c_longlong]
log.restype = C.c_int
out = []
for rep in range(5):
    B.check(log(None, nwaves))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); step(); ev1.record()
    torch.cuda.synchronize()
    buf = np.zeros((nwaves, 4), dtype=np.uint64)
    B.check(log(buf.ctypes.data, nwaves))
    ok = buf[:, 1] > 0
    t0 = buf[ok, 0].astype(np.int64); t1 = buf[ok, 1].astype(np.int64)
    base = t0.min()
    start = (t0 - base) / 100.0; end = (t1 - base) / 100.0; dur = end - start
    hw = buf[ok, 2].astype(np.int64); xcc = buf[ok, 3].astype(np.int64) & 0xf
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    slot = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd  # one SIMD
    uniq, inv, cnt = np.unique(slot, return_inverse=True, return_counts=True)
    share = cnt[inv]  # waves of this launch on the same SIMD
    q = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 99, 100)]  # noqa: E731
    order = np.argsort(start)
    # rounds: for every SIMD, its waves in start order -> (start, end) of the 1st, 2nd, 3rd ... wave it ran
    per_simd = {}
    for i in order:
        per_simd.setdefault(int(slot[i]), []).append((float(start[i]), float(end[i])))
    nth = {}
    for v in per_simd.values():
        for k, (a_, b_) in enumerate(v):
            nth.setdefault(k, []).append((a_, b_, b_ - a_))
    # the pacing rule's prediction (cloudsc2_kernels.hip: set_pace; one wave per SIMD: 2 waves per workgroup, CUs x 2 slots): workgroups
    # at position p < rem of a round are expected on the slots that run one workgroup more -- checked against where they really ran
    pace = None
    if kind != "nl":
        idx = np.nonzero(ok)[0]
        wg = idx // 2
        slots = 2 * torch.cuda.get_device_properties(0).multi_processor_count
        nwg = nwaves // 2
        k_rounds, rem = nwg // slots, nwg % slots
        fast = (wg % slots) < rem
        pace = {"slots": slots, "whole_rounds": int(k_rounds), "partial_round": int(rem),
                "fast_class_waves_by_waves_on_their_simd": {int(k): int(((share == k) & fast).sum()) for k in np.unique(share)},
                "slow_class_waves_by_waves_on_their_simd": {int(k): int(((share == k) & ~fast).sum()) for k in np.unique(share)},
                "end_us_fast_class_p50_p100": [round(float(np.median(end[fast])), 1), round(float(end[fast].max()), 1)] if fast.any() else None,
                "end_us_slow_class_p50_p100": [round(float(np.median(end[~fast])), 1), round(float(end[~fast].max()), 1)] if (~fast).any() else None}
    if kind == "nl":  # the population rule behind the light-SIMD nap (cloudsc2_simd_population) against where the waves really ran
        idx = np.nonzero(ok)[0]
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        a_, b_ = C.c_int(), C.c_int()
        pred = np.zeros(len(idx), dtype=np.int64)
        for n_, w_ in enumerate(idx):
            B.check(B.lib.cloudsc2_simd_population(nwaves // 2, cus, int(w_) // 2, int(w_) & 1, C.byref(a_), C.byref(b_)))
            pred[n_] = a_.value
        pace = {"rule": "cloudsc2_simd_population", "waves": int(len(idx)), "predicted_population_equals_waves_on_the_simd": int((pred == share).sum()),
                "end_us_median_by_waves_on_the_simd": {int(k): round(float(np.median(end[share == k])), 1) for k in np.unique(share)}}
    r = {"event_ms": ev0.elapsed_time(ev1), "kernel": kind, "pace_rule_vs_reality": pace,
         "nth_wave_of_its_simd_start_end_duration_us_median": {k: [round(float(np.median([x[j] for x in v])), 1) for j in range(3)] + [len(v)] for k, v in sorted(nth.items()) if k < 8}, "waves": int(ok.sum()), "simds_used": int(len(uniq)),
         "start_us_p0_10_50_90_99_100": q(start), "end_us": q(end), "duration_us": q(dur),
         "waves_per_simd_hist": {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))},
         "end_us_median_by_share": {int(k): round(float(np.median(end[share == k])), 1) for k in np.unique(share)},
         "duration_us_median_by_share": {int(k): round(float(np.median(dur[share == k])), 1) for k in np.unique(share)},
         "end_us_median_by_xcd": {int(k): round(float(np.median(end[xcc == k])), 1) for k in np.unique(xcc)},
         "waves_by_xcd": {int(k): int((xcc == k).sum()) for k in np.unique(xcc)}}
    out.append(r)
    if os.environ.get("WAVE_TIMES_RAW"):
        np.save(os.environ["WAVE_TIMES_RAW"] + f"_{ngptot}_{rep}.npy", buf)
print(json.dumps({"ngptot": ngptot, "nproma": nproma, "launches": out}, indent=1))
