#!/bin/bash
# A/B of library builds on the evaporation-branch NL sweep: tools/ab_evap.sh TAG "libA.so libB.so" [rounds]
tag=$1; libs=$2; rounds=${3:-3}
out=gpurun_out/$tag; mkdir -p $out
for r in $(seq $rounds); do for n in 160000 1048576; do for lib in $libs; do
  steps=200; [ $n = 1048576 ] && steps=50
  CLOUDSC2_LIB=$lib python bench.py --kernel nl --levapls2 --ngptot $n --steps $steps --warmup 5 --no-cpu-baseline --no-companions 2>>$out/err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$n', '$lib', d['roofline']['kernel_ms_avg'])" >> $out/ab.txt || exit 1
done; done; done
sort $out/ab.txt
