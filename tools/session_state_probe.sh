#!/bin/bash
# Placement judged by the NL sweep itself (cloudsc2_device_malloc_state, the default for states) against the generic two-stream
# ranking (CLOUDSC2_PLACE_PROBE=both is not a keyword: any value other than "kernel" switches the kernel probe off; "both" here =
# CLOUDSC2_PLACE_PROBE=generic).  Fresh process per line, alternating.  usage: tools/session_state_probe.sh TAG [ITER] [SIZES] [KERNELS]
tag=${1:-sp}; it=${2:-4}; sizes=${3:-"160000 1048576"}; kernels=${4:-nl}; out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do for i in $(seq 1 $it); do for pr in kernel generic; do for k in $kernels; do
  CLOUDSC2_PLACE_PROBE=$pr CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel $k --ngptot $n --steps 50 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$pr $k $n kernel ms avg', round(r['kernel_ms_avg'],4), 'frac', round(r['frac'],4))"
  grep "cloudsc2_device_malloc" $out/err.log | cut -c1-560
done; done; done; done | tee $out/summary.txt
