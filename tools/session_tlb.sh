#!/bin/bash
# Is the slow write class an address-translation effect?  TCP_UTCL1_* and TCP write counters of the NL-shaped write stream on the
# fastest (nl_writes<2>) and the slowest (nl_writes<1>) of 60 separate 1.4 GB allocations (tools/hbm_probe pmc 60), one rocprofv3
# --pmc pass per counter set (csv; per-dispatch durations come from the same csv).  usage: tools/session_tlb.sh TAG
tag=${1:-r02_w}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $out/avail.txt 2>&1 || true
grep -i -o "UTCL2[A-Z0-9_]*\|[A-Z0-9_]*TLB[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*LATENCY[A-Z0-9_]*\|TCP_[A-Z0-9_]*WRITE[A-Z0-9_]*" $out/avail.txt | sort -u > $out/avail_names.txt
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d /tmp/tlb_$i -- /root/repo/tools/bin/hbm_probe pmc 60 > $out/pass_$i.log 2>&1 || { echo "pass $i failed ($set)"; grep -m2 "rror\|F2026" $out/pass_$i.log | cut -c1-200; continue; }
  grep "^fastest" $out/pass_$i.log
  python3 - /tmp/tlb_$i <<'PY'
import csv, glob, os, sys
acc = {}
dur = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = "FAST" if "nl_writes<2" in row["Kernel_Name"] else "SLOW" if "nl_writes<1" in row["Kernel_Name"] else None
        if not k: continue
        acc.setdefault((k, row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
        dur.setdefault(k, {})[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
for k in sorted(dur):
    v = sorted(dur[k].values())
    print(f"{k} dispatches {len(v)} duration ms median {v[len(v)//2]:.4f} min {v[0]:.4f}")
for (k, c) in sorted(acc, key=lambda x: (x[1], x[0])):
    v = acc[(k, c)]
    print(f"{k} {c:48s} {sum(v.values()) / len(v):16.1f}")
PY
done | tee $out/summary.txt
cat $out/avail_names.txt | tr '\n' ' '
