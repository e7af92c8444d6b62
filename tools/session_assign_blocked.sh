#!/bin/bash
# AD in its assign form (bench.py --kernel ad --ad-assign) next to the accumulate form, and the interleaved ("blocked") scratch
# layout of the perturbation sets at NPROMA 64 / 128, where the library uses the flat one.  usage: tools/session_assign_blocked.sh TAG
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
line() { python -c "import json,sys; d=json.load(open('$1')); print('$2', round(d['roofline']['kernel_ms_avg'],4), 'ms', round(d['roofline']['frac'],4), d['roofline']['bytes_per_column'], 'B/col')"; }
for n in 160000 1048576; do for form in "" "--ad-assign"; do
  f=$out/ad_${n}_${form:-accumulate}.json
  timeout -k 10 200 python bench.py --kernel ad $form --ngptot $n --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $f 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  line $f "ad $n ${form:-accumulate}"
done; done | tee $out/summary.txt
for p in 64 128; do for k in tl ad; do for lay in flat blocked; do
  f=$out/${k}_nproma${p}_$lay.json
  CLOUDSC2_SCRATCH_LAYOUT=$lay timeout -k 10 200 python bench.py --kernel $k --nproma $p --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $f 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  line $f "$k nproma $p $lay"
done; done; done | tee -a $out/summary.txt
