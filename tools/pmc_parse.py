#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_workload.py into per-launch HBM traffic.

    python tools/pmc_parse.py gpurun_out/pmc_fetch gpurun_out/pmc_write NGPTOT > profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024).  gfx950 under-reports wide
coalesced reads by exactly 2x for 16 B per lane; this code reads 8 B per lane, so the read and write factors are
calibrated on the SATUR dispatch whose byte count is known (2 planes in, 1 plane out)."""
import csv
import glob
import re
import json
import os
import sys


def per_kernel(dirname, counter):
    acc = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            m = re.search(r"nl_kernel<(\d+)u?>", name)
            if "ad_reverse_kernel" in name:
                key = "ad_rev"   # two-kernel form of the adjoint: reverse pass ...
            elif m and int(m.group(1)) & 16:
                key = "ad_fwd"   # ... and its trajectory pass (nl_kernel with C2F_CKPT)
            else:
                key = ("satur" if "satur_kernel" in name else "nl" if "nl_kernel" in name else "tl" if "tl_kernel" in name
                       else "ad" if "ad_kernel" in name else None)
            if key is None:
                continue
            acc.setdefault(key, {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[key][row["Dispatch_Id"]] += float(row["Counter_Value"])
    avg = {k: sum(v.values()) / len(v) for k, v in acc.items()}
    cnt = {k: len(v) for k, v in acc.items()}
    if "ad" not in avg and "ad_fwd" in avg and "ad_rev" in avg:  # one adjoint launch = one of each
        avg["ad"] = avg.pop("ad_fwd") + avg.pop("ad_rev")
        cnt["ad"] = cnt.pop("ad_rev")
        cnt.pop("ad_fwd")
    return avg, cnt


def main():
    fetch_dir, write_dir, ngptot = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rb = int(sys.argv[4]) if len(sys.argv) > 4 else 8  # bytes per real: 4 for the fp32 library (CLOUDSC2_PRECISION=single)
    nlev = 137
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(write_dir, "WRITE_SIZE")
    plane = ngptot * nlev * rb
    cal_r = (2 * plane) / (fetch["satur"] * 1024)
    cal_w = (1 * plane) / (write["satur"] * 1024)
    algo = {"nl": 28536 * rb // 8, "tl": 57072 * rb // 8, "ad": (85608 + 2 * 8 * nlev) * rb // 8}  # bytes per column, DESIGN.md
    out = {"ngptot": ngptot, "real_bytes": rb, "unit": "bytes per launch", "calibration": {"kernel": f"satur_kernel ({rb} B/lane, 2 planes in, 1 out)",
           "read_factor": cal_r, "write_factor": cal_w, "raw_fetch_kib": fetch["satur"], "raw_write_kib": write["satur"]},
           "dispatches": {"fetch": nf, "write": nw}, "kernels": {}}
    for k in ("nl", "tl", "ad"):
        if k in fetch and k in write:
            rd = fetch[k] * 1024 * cal_r
            wr = write[k] * 1024 * cal_w
            out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr,
                                 "algorithmic_bytes": algo[k] * ngptot, "traffic_over_algorithmic": (rd + wr) / (algo[k] * ngptot),
                                 "raw_fetch_kib": fetch[k], "raw_write_kib": write[k]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
