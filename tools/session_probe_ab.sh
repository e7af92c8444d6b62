#!/bin/bash
# A/B of how the allocator ranks whole-size candidates: by the sweeps' write stream ("write", round 2's first form), by the NL sweep's
# whole pattern ("mixed"), or by both (the default, "both").  Fresh process per line, alternating; the allocator's per-candidate probe
# times follow each line.  usage: tools/session_probe_ab.sh TAG [ITER] [KERNELS]
tag=${1:-probe_ab}; it=${2:-5}; kernels=${3:-"nl tl"}; out=gpurun_out/$tag; mkdir -p $out
run() {  # probe kernel ngptot
  if [ $1 = both ]; then unset CLOUDSC2_PLACE_PROBE; else export CLOUDSC2_PLACE_PROBE=$1; fi
  CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel $2 --ngptot $3 --steps ${STEPS:-100} --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('$1 $2 $3 kernel ms avg', round(r['kernel_ms_avg'],4), 'first/last tenth', round(r['kernel_ms_first_tenth'],4), round(r['kernel_ms_last_tenth'],4))"
  grep "cloudsc2_device_malloc:" $out/err.log | cut -c1-420
}
for i in $(seq 1 $it); do for pr in both write mixed; do for k in $kernels; do run $pr $k 160000; done; done; done | tee $out/summary.txt
