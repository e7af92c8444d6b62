#!/bin/bash
# 1 048 576 columns, placement judged by the NL sweep: whole-size hipMalloc candidates (CLOUDSC2_PLACE_MODE=whole, 4 fit) against the
# composed form (VMM chunks).  Fresh process per line, alternating.  usage: tools/session_mode_ab.sh TAG [ITER] [KERNELS]
tag=${1:-mode_ab}; it=${2:-5}; kernels=${3:-nl}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $it); do for mode in whole chunks; do for k in $kernels; do
  CLOUDSC2_PLACE_MODE=$mode CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel $k --ngptot 1048576 --steps 50 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$mode $k 1048576 kernel ms avg', round(r['kernel_ms_avg'],4), 'frac', round(r['frac'],4))"
  grep "cloudsc2_device_malloc" $out/err.log | cut -c1-400
done; done; done | tee $out/summary.txt
