#!/bin/bash
# The final allocator (whole-size hipMalloc candidates judged by the NL sweep for states): fresh process per line, all three kernels at
# both sizes.  usage: tools/session_final_placement.sh TAG [ITER]
tag=${1:-fp}; it=${2:-4}; out=gpurun_out/$tag; mkdir -p $out
for n in ${SIZES:-160000 1048576}; do for i in $(seq 1 $it); do for k in ${KERNELS:-nl tl ad}; do
  CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel $k --ngptot $n --steps 50 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$k $n kernel ms avg', round(r['kernel_ms_avg'],4), 'frac', round(r['frac'],4))"
  grep "cloudsc2_device_malloc" $out/err.log | cut -c1-400
done; done; done | tee $out/summary.txt
