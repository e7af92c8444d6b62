#!/bin/bash
# A/B of library builds through bench.py, fresh process per line, alternating, first allocation placed by the library.
# usage: LIBS="real= skeleton=tools/bin/libcloudsc2_hip_skel.so" tools/session_lib_ab.sh TAG [ITER] [SIZES] [KERNELS]
tag=${1:-lib_ab}; it=${2:-4}; sizes=${3:-"160000 1048576"}; kernels=${4:-nl}; out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do for i in $(seq 1 $it); do for k in $kernels; do for spec in ${LIBS:-"real="}; do
  name=${spec%%=*}; path=${spec#*=}
  if [ -n "$path" ]; then export CLOUDSC2_LIB=$PWD/$path; else unset CLOUDSC2_LIB; fi
  timeout -k 10 300 python bench.py --kernel $k --ngptot $n --steps 100 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$name $k $n kernel ms avg', round(r['kernel_ms_avg'],4), 'min', round(r['kernel_ms_min'],4), 'frac', round(r['frac'],4))"
done; done; done; done | tee $out/summary.txt
