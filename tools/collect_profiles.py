#!/usr/bin/env python3
"""Copy the judged summaries of an evidence session (tools/evidence.sh / evidence_sp.sh) from gpurun_out/TAG into
profiles/ under PREFIX:   python tools/collect_profiles.py gpurun_out/r01_k r01_k"""
import glob
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
cp = lambda a, b: shutil.copy(os.path.join(src, a), os.path.join(dst, f"{prefix}_{b}"))  # noqa: E731
for n in (160000, 1048576):
    cp(f"pmc_traffic_{n}.json", f"{n}_pmc_traffic.json")
cp("bench_default.json", "bench_default.json")
cp("bench_prof.json", "bench_under_rocprof.json")
cp("bench_prof_timed_region.json", "bench_under_rocprof_timed_region.json")
shutil.copy(glob.glob(os.path.join(src, "prof", "*", "*kernel_stats.csv"))[0], os.path.join(dst, f"{prefix}_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_prof_1m.json")):  # the north_star target configuration under the profiler
    cp("bench_prof_1m.json", "bench_1m_under_rocprof.json")
    cp("bench_prof_1m_timed_region.json", "bench_1m_under_rocprof_timed_region.json")
    shutil.copy(glob.glob(os.path.join(src, "prof_1m", "*", "*kernel_stats.csv"))[0], os.path.join(dst, f"{prefix}_bench_1m_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "timed_kernel_stats.csv")):  # one row per timed kernel (NL, TL, AD, AD reverse, Taylor) and size
    cp("timed_kernel_stats.csv", "timed_kernel_stats.csv")
    for d in sorted(glob.glob(os.path.join(src, "prof_*_*"))):  # rocprofv3's own --stats summaries of those runs
        for f in glob.glob(os.path.join(d, "*", "*kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{prefix}_{os.path.basename(d)[5:]}_kernel_stats.csv"))
    for f in glob.glob(os.path.join(src, "bench_prof_*_*.json")):
        name = os.path.basename(f)[len("bench_prof_"):-len(".json")]
        if name.split("_")[0] in ("tl", "ad", "adrev", "selftests", "nlevap"):  # (the NL runs are copied above under their older names)
            shutil.copy(f, os.path.join(dst, f"{prefix}_{name}_under_rocprof.json"))
if os.path.exists(os.path.join(src, "pytest_gpu.log")):
    cp("pytest_gpu.log", "pytest_gpu.log")
for opt in ("rocm_smi_during_bench.txt", "single_checks_gpu.log"):
    if os.path.exists(os.path.join(src, opt)):
        cp(opt, opt)
raw = os.path.join(dst, f"{prefix}_pmc_raw")
os.makedirs(raw, exist_ok=True)
for n in (160000, 1048576):
    for w in ("fetch", "write"):
        shutil.copy(glob.glob(os.path.join(src, f"pmc_{w}_{n}", "*", "*counter_collection.csv"))[0],
                    os.path.join(raw, f"{w}_{n}_counter_collection.csv"))
out = {}
for k in ("nl", "tl", "ad"):
    for n in (160000, 1048576):
        d = json.load(open(os.path.join(src, f"bench_{k}_{n}.json")))
        out[f"{k}_{n}"] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "dtype": d["dtype"], "roofline": d["roofline"],
                           "placement": d["config"]["placement"]}
        print(prefix, k, n, round(d["roofline"]["kernel_ms_avg"], 3), "ms", round(100 * d["roofline"]["frac"], 1), "%", "%.3e" % d["value"])
for form in ("adassign", "adreverse", "adreverseassign"):  # the adjoint's other forms (tools/evidence.sh)
    for n in (16384, 160000, 1048576):
        f = os.path.join(src, f"bench_{form}_{n}.json")
        if os.path.exists(f) and os.path.getsize(f):
            d = json.load(open(f))
            out[f"{form}_{n}"] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "dtype": d["dtype"], "roofline": d["roofline"]}
for n in (16384, 160000, 1048576):
    f = os.path.join(src, f"symmetry_{n}.json")
    if os.path.exists(f) and os.path.getsize(f):
        out[f"adjoint_test_{n}"] = json.load(open(f))
f = os.path.join(src, "taylor_test.jsonl")
if os.path.exists(f) and os.path.getsize(f):
    out["taylor_test"] = [json.loads(line) for line in open(f) if line.strip()]
json.dump(out, open(os.path.join(dst, f"{prefix}_bench_all_kernels.json"), "w"), indent=1)
