#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_workload.py into per-launch HBM traffic.

    python tools/pmc_parse.py gpurun_out/pmc_fetch gpurun_out/pmc_write NGPTOT [REAL_BYTES] > profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024).  gfx950 under-reports wide
coalesced reads by exactly 2x for 16 B per lane; this code reads 8 B per lane, so the read and write factors are
calibrated on the SATUR dispatch whose byte count is known (2 planes in, 1 plane out).  The dispatches are attributed to the
launches of tools/pmc_plan.py by their ORDER (a launch of the adjoint is one fused kernel or a forward + a reverse kernel)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_plan import PLAN  # noqa: E402

SWEEPS = ("satur_kernel", "nl_kernel", "tl_kernel", "ad_reverse_kernel", "ad_kernel")


def sweep_of(name):
    for k in SWEEPS:
        if k in name:
            return k
    return None


def per_launch(dirname, counter):
    """{label: (average counter value per launch, launches)} -- dispatches of the sweep kernels in dispatch order vs the plan"""
    disp = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter or sweep_of(row["Kernel_Name"]) is None:
                continue
            d = disp.setdefault(int(row["Dispatch_Id"]), [sweep_of(row["Kernel_Name"]), 0.0])
            d[1] += float(row["Counter_Value"])  # one row per XCD / dimension: summed
    seq = [disp[k] for k in sorted(disp)]
    out, pos = {}, 0
    for label, reps in PLAN:
        vals = []
        for _ in range(reps):
            if pos >= len(seq):
                raise SystemExit(f"{dirname}: the profile ends before the plan does (at {label})")
            kern, v = seq[pos]
            pos += 1
            want = {"satur": "satur_kernel", "nl": "nl_kernel", "tl": "tl_kernel"}.get(label)
            if label.startswith("ad_reverse"):
                want = "ad_reverse_kernel"
            if want is None and kern == "nl_kernel":  # the whole adjoint as two kernels: forward (the NL sweep) + reverse
                k2, v2 = seq[pos]
                pos += 1
                if k2 != "ad_reverse_kernel":
                    raise SystemExit(f"{dirname}: {label}: nl_kernel not followed by ad_reverse_kernel")
                v += v2
            elif want is None and kern != "ad_kernel":
                raise SystemExit(f"{dirname}: {label}: unexpected {kern}")
            elif want is not None and kern != want:
                raise SystemExit(f"{dirname}: {label}: expected {want}, found {kern}")
            vals.append(v)
        out[label] = (sum(vals) / len(vals), len(vals))
    if pos != len(seq):
        raise SystemExit(f"{dirname}: {len(seq) - pos} sweep dispatches beyond the plan (was CLOUDSC2_PLACE=0 set?)")
    return out


def traffic(fetch_dir, write_dir, ngptot, rb=8):
    """The parsed result as a dict (bench.py measures the traffic in its own run through this)."""
    nlev = 137
    from dwarf_p_cloudsc2_tl_ad_amd.state import bytes_per_column as bpc

    fetch = per_launch(fetch_dir, "FETCH_SIZE")
    write = per_launch(write_dir, "WRITE_SIZE")
    plane = ngptot * nlev * rb
    cal_r = (2 * plane) / (fetch["satur"][0] * 1024)
    cal_w = (1 * plane) / (write["satur"][0] * 1024)
    old = bpc(nlev, "ad_old_adjoints", rb)
    algo = {"nl": bpc(nlev, "nl_driver", rb), "tl": bpc(nlev, "tl", rb), "ad": bpc(nlev, "ad", rb), "ad_assign": bpc(nlev, "ad", rb) - old,
            "ad_reverse": bpc(nlev, "ad_reverse", rb), "ad_reverse_assign": bpc(nlev, "ad_reverse", rb) - old}  # bytes per column, DESIGN.md
    out = {"ngptot": ngptot, "real_bytes": rb, "unit": "bytes per launch",
           "calibration": {"kernel": f"satur_kernel ({rb} B/lane, 2 planes in, 1 out)", "read_factor": cal_r, "write_factor": cal_w,
                           "raw_fetch_kib": fetch["satur"][0], "raw_write_kib": write["satur"][0]},
           "dispatches": {"fetch": {k: v[1] for k, v in fetch.items()}, "write": {k: v[1] for k, v in write.items()}}, "kernels": {}}
    for k in algo:
        rd = fetch[k][0] * 1024 * cal_r
        wr = write[k][0] * 1024 * cal_w
        out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr, "algorithmic_bytes": algo[k] * ngptot,
                             "traffic_over_algorithmic": (rd + wr) / (algo[k] * ngptot), "raw_fetch_kib": fetch[k][0],
                             "raw_write_kib": write[k][0]}
    return out


def main():
    fetch_dir, write_dir, ngptot = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rb = int(sys.argv[4]) if len(sys.argv) > 4 else 8  # bytes per real: 4 for the fp32 library (CLOUDSC2_PRECISION=single)
    print(json.dumps(traffic(fetch_dir, write_dir, ngptot, rb), indent=1))


if __name__ == "__main__":
    main()
