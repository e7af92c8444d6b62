#!/bin/bash
# Fresh processes of the NL bench, 300 steps: both probes of every candidate (stderr of the allocator) and the kernel's time in the
# first and the last tenth of the timed region -- is a slow run slow throughout, and what did the probes say about the chosen candidate?
tag=${1:-drift}; n=${2:-8}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do for pr in mixed write; do
  CLOUDSC2_PLACE_PROBE=$pr CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel nl --steps 300 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$pr nl avg', round(r['kernel_ms_avg'],4), 'min', round(r['kernel_ms_min'],4), 'first tenth', round(r['kernel_ms_first_tenth'],4), 'last tenth', round(r['kernel_ms_last_tenth'],4))"
  grep "cloudsc2_device_malloc:" $out/err.log | cut -c1-400
done; done | tee $out/summary.txt
