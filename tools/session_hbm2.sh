#!/bin/bash
# same box, three views: separate allocations (hbm_probe), one big allocation (hbm_map), many allocations (hbm_alloc)
tag=${1:-r02_d}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 tools/bin/hbm_probe map 150 > $out/probe_map.txt 2>&1 || { echo probe failed; tail -3 $out/probe_map.txt; exit 1; }
python3 - $out/probe_map.txt <<'PY'
import sys
for ln in open(sys.argv[1]):
    if ln.startswith("map round 1"):
        v = [float(x) for x in ln.split(":")[1].split()]
        print("probe (150 separate 1.4 GB allocations), NL-shaped write ms, class map (F < 0.165 <= S):")
        print("".join("F" if x < 0.165 else "S" for x in v))
        print("min %.4f max %.4f" % (min(v), max(v)))
    if ln.startswith("fastest"): print(ln.strip())
PY
timeout -k 10 300 tools/bin/hbm_map map 250 2048 > $out/map_2048.txt 2>&1 || { echo map failed; tail -3 $out/map_2048.txt; exit 1; }
cat $out/map_2048.txt
timeout -k 10 300 tools/bin/hbm_alloc 150 A > $out/alloc_150.txt 2>&1 || { echo alloc failed; tail -3 $out/alloc_150.txt; exit 1; }
python3 - $out/alloc_150.txt <<'PY'
import sys
v = [float(ln.split()[-1]) for ln in open(sys.argv[1]) if ln.startswith("  0x")]
print("hbm_alloc A x150: NL-shaped GB/s min %.0f max %.0f; sorted deciles:" % (min(v), max(v)), [round(sorted(v)[i * len(v) // 10]) for i in range(10)])
print("".join("F" if x > 0.5 * (min(v) + max(v)) else "S" for x in v))
PY
i=0
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_BUSY TCC_TAG_STALL" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_64B TCC_REQ"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --output-format json -d /tmp/pmc_$i -- $OLDPWD/tools/bin/hbm_probe pmc 60 > $OLDPWD/$out/pmc_$i.log 2>&1 ) || { echo "pmc set $i failed"; tail -3 $out/pmc_$i.log; continue; }
  grep "^fastest" $out/pmc_$i.log
  f=$(ls /tmp/pmc_$i/*/*_results.json | head -1)
  python3 tools/pmc_json_summary.py $f 'nl_writes<[12],' > $out/pmc_summary_$i.json 2> $out/pmc_summary_$i.err || { echo "summary $i failed"; tail -3 $out/pmc_summary_$i.err; }
  python3 - $out/pmc_summary_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("kernels", {}).items():
    for c, s in v["counters"].items():
        print(f"{k[:24]:24s} {c:34s} n={s['instances']:4d} sum={s['sum']:14.0f} min={s['min']:10.0f} max={s['max']:10.0f}")
PY
done
