#!/bin/bash
# same box, fresh processes, alternating: whole state in ONE placed allocation (one) / read-only and written arrays in two placed
# allocations (both) / written placed, read-only plain (split) / no placement (none):  tools/first_alloc_ab.sh OUT N KERNEL NGPTOT
out=${1:-gpurun_out/first_ab.txt}; n=${2:-5}; k=${3:-nl}; g=${4:-160000}
: > $out
for i in $(seq 1 $n); do
  CLOUDSC2_STATE_SPLIT=0 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/one   /' >> $out
  CLOUDSC2_STATE_SPLIT=2 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/both  /' >> $out
  CLOUDSC2_STATE_SPLIT=1 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/split /' >> $out
  CLOUDSC2_PLACE=0 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/none  /' >> $out
done
python3 - $out <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    tag, js = ln.split(None, 1); d = json.loads(js)
    print(tag, d["kernel"], d["ngptot"], d["ms_median"], d["placement"].get("candidates"), round(d["placement"].get("probe_ms_best", 0), 4), round(d["placement"].get("probe_ms_median", 0), 4))
PY
