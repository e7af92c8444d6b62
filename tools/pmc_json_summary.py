#!/usr/bin/env python3
"""Per-instance summary of a rocprofv3 --pmc run written with --output-format json: for every kernel name that matches
KERNEL_RE and every counter, the per-instance values (e.g. the 128 TCC channels) averaged over the kernel's dispatches.
    python tools/pmc_json_summary.py RESULTS.json KERNEL_RE > summary.json
The raw JSON (tens of MB) can then be deleted; only the summary travels back from the GPU box."""
import collections
import json
import re
import sys


def main():
    path, kre = sys.argv[1], re.compile(sys.argv[2])
    doc = json.load(open(path))
    tool = doc["rocprofiler-sdk-tool"]
    tool = tool[0] if isinstance(tool, list) else tool
    out = {"file": path, "schema": {}}
    # id -> name tables
    counters = {}
    for c in tool.get("counters", []):
        cid = c.get("id", {})
        counters[cid.get("handle", cid) if isinstance(cid, dict) else cid] = c.get("name")
    ksyms = {}
    for k in tool.get("kernel_symbols", []):
        ksyms[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name") or k.get("truncated_kernel_name")
    recs = tool.get("callback_records", {}).get("counter_collection", [])
    out["schema"]["n_records"] = len(recs)
    if recs:
        out["schema"]["record_example"] = json.loads(json.dumps(recs[0]))  # one record verbatim
        if isinstance(out["schema"]["record_example"].get("records"), list):
            out["schema"]["record_example"]["records"] = out["schema"]["record_example"]["records"][:6]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    ndisp = collections.Counter()
    for r in recs:
        dd = r.get("dispatch_data", {})
        info = dd.get("dispatch_info", dd)
        kname = ksyms.get(info.get("kernel_id"), str(info.get("kernel_id")))
        if not kre.search(kname or ""):
            continue
        ndisp[kname] += 1
        per_counter = collections.defaultdict(list)
        for v in r.get("records", []):
            cid = v.get("counter_id", {})
            cid = cid.get("handle", cid) if isinstance(cid, dict) else cid
            per_counter[counters.get(cid, str(cid))].append(v.get("value"))
        for cname, vals in per_counter.items():
            for i, x in enumerate(vals):
                agg[kname][cname][i].append(x)
    out["kernels"] = {}
    for kname, cs in agg.items():
        ko = {"dispatches": ndisp[kname], "counters": {}}
        for cname, inst in cs.items():
            means = [sum(v) / len(v) for _, v in sorted(inst.items())]
            ko["counters"][cname] = {"instances": len(means), "sum": sum(means), "min": min(means), "max": max(means),
                                     "per_instance_mean": [round(m, 1) for m in means]}
        out["kernels"][kname] = ko
    json.dump(out, sys.stdout, indent=None)
    print()


if __name__ == "__main__":
    main()
