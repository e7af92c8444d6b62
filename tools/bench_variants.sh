#!/bin/bash
# usage: tools/bench_variants.sh "lib1.so lib2.so ..." "kernels" "ngptots"
for lib in $1; do for k in $2; do for n in $3; do
  CLOUDSC2_LIB=$lib timeout -k 10 200 python bench.py --kernel $k --ngptot $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['metric'][:12], d['config']['ngptot_per_gpu'], round(d['ms_per_step'],3), '%.3e'%d['value'], round(d['roofline']['frac'],3))"
done; done; done
