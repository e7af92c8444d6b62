#!/bin/bash
# TL / AD: perturbation set inside the state's allocation (ONE_ARENA=1) vs in its own placed allocation (0); fresh processes, alternating
out=$1; n=$2; k=$3; g=${4:-160000}
: > $out
for i in $(seq 1 $n); do
  ONE_ARENA=1 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/one      /' >> $out
  ONE_ARENA=0 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 | sed 's/^/separate /' >> $out
done
python3 - $out <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    tag, js = ln.split(None, 1); d = json.loads(js)
    print(tag, d["kernel"], d["ngptot"], d["ms_median"], d["placement"].get("candidates"))
PY
