#!/bin/bash
# A/B of the allocator's candidate score: whole candidate's probe time ("total", round 2's first form) against the slowest ~1 GiB
# segment ("worst").  Fresh process per line, alternating.  usage: tools/session_score.sh TAG [ITER]
tag=${1:-score}; it=${2:-4}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $it); do for sc in total worst; do
  for k in nl tl ad; do
    CLOUDSC2_PLACE_SCORE=$sc timeout -k 10 200 python bench.py --kernel $k --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
    python -c "import json; d=json.load(open('$out/b.json')); p=d['config']['placement']; print('$sc $k', round(d['roofline']['kernel_ms_avg'],4), '|', p.get('candidates'), p.get('probe_best_ms'), p.get('probe_median_ms'), p.get('probe_worst_ms'))"
  done
  CLOUDSC2_PLACE_SCORE=$sc timeout -k 10 100 tools/bin/hbm_width 160000 1 2>&1 | grep shaped | tail -1 | sed "s/^/$sc skeleton /"
done; done | tee $out/summary.txt
