#!/bin/bash
# Interleaved A/B of library builds over kernels and sizes, fresh process per run, placed states:
#   tools/ab_libs.sh TAG "libA.so libB.so" "nl tl ad" "160000 1048576" [rounds]
tag=$1; libs=$2; kernels=$3; sizes=$4; rounds=${5:-3}
out=gpurun_out/$tag; mkdir -p $out
for r in $(seq $rounds); do for k in $kernels; do for n in $sizes; do for lib in $libs; do
  steps=200; [ $n -gt 500000 ] && steps=40
  CLOUDSC2_LIB=$lib python bench.py --kernel $k --ngptot $n --steps $steps --warmup 5 --no-cpu-baseline --no-companions 2>>$out/err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$k', '$n', '$(basename $lib)', round(d['roofline']['kernel_ms_avg'], 4))" >> $out/ab.txt || exit 1
done; done; done; done
sort $out/ab.txt
