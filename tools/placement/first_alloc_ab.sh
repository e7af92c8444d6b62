#!/bin/bash
# same box, fresh processes, alternating placement schemes of cloudsc2_device_malloc:
#   whole  = whole-size hipMalloc candidates (default), chunks = composed of probed 2 GiB hipMemCreate chunks, none = no placement
out=${1:-gpurun_out/first_ab.txt}; n=${2:-5}; k=${3:-nl}; g=${4:-160000}
: > $out
for i in $(seq 1 $n); do
  timeout -k 10 200 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/whole  /' >> $out
  CLOUDSC2_PLACE_MODE=chunks timeout -k 10 200 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/chunks /' >> $out
  CLOUDSC2_PLACE=0 timeout -k 10 200 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/none   /' >> $out
done
python3 - $out <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    tag, js = ln.split(None, 1); d = json.loads(js)
    print(tag, d["kernel"], d["ngptot"], d["ms_median"], d["placement"].get("candidates"), round(d["placement"].get("probe_ms_best", 0), 4), round(d["placement"].get("probe_ms_median", 0), 4), round(d["placement"].get("probe_ms_worst", 0), 4))
PY
