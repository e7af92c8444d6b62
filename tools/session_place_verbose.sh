#!/bin/bash
# N fresh processes of the default NL bench with the allocator's per-candidate probe times on stderr: when a placement misses, was no
# fast candidate found or did the probe mispredict?  usage: tools/session_place_verbose.sh TAG [N]
tag=${1:-pv}; n=${2:-8}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do for sc in total worst; do
  CLOUDSC2_PLACE_SCORE=$sc CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 200 python bench.py --kernel nl --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); print('$sc nl', round(d['roofline']['kernel_ms_avg'],4))"
  grep "cloudsc2_device_malloc:" $out/err.log
done; done | tee $out/summary.txt
