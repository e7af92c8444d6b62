#!/bin/bash
# A/B of the composed placement (states above 12 GiB) at 1 048 576 columns: "greedy" = the chunks that are fastest one by one (round 2's
# first form) against the default, which times whole compositions (greedy + windows of consecutive chunks) with both probes.
# Fresh process per line, alternating.  usage: tools/session_compose_ab.sh TAG [ITER] [KERNELS]
tag=${1:-compose_ab}; it=${2:-4}; kernels=${3:-nl}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $it); do for how in windows greedy; do for k in $kernels; do
  if [ $how = greedy ]; then export CLOUDSC2_PLACE_COMPOSE=greedy; else unset CLOUDSC2_PLACE_COMPOSE; fi
  t0=$(date +%s.%N)
  CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel $k --ngptot 1048576 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  t1=$(date +%s.%N)
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$how $k 1048576 kernel ms avg', round(r['kernel_ms_avg'],4), 'frac', round(r['frac'],4), '| process wall s', round($t1-$t0,1))"
  grep "cloudsc2_device_malloc" $out/err.log | cut -c1-600
done; done; done | tee $out/summary.txt
