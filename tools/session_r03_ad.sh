#!/bin/bash
# Round 3, AD traffic work: GPU suite, the adjoint in its four forms at both sizes, the symmetry driver's timing, PMC traffic passes.
# usage: tools/session_r03_ad.sh TAG
tag=${1:-r03_a}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for n in 160000 1048576; do
  for form in "" "--ad-assign" "--ad-sweep reverse" "--ad-sweep reverse --ad-assign"; do
    f=$out/bench_ad_${n}_$(echo $form | tr -d ' -').json
    timeout -k 10 300 python bench.py --kernel ad $form --ngptot $n --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $f 2> $f.err || { tail -3 $f.err; exit 1; }
    python -c "import json; d=json.load(open('$f')); r=d['roofline']; print('ad $n [$form]', round(r['kernel_ms_avg'],3), 'ms', r['bytes_per_column'], 'B/col', round(r['frac'],3))"
  done
done
for n in 16384 160000; do
  timeout -k 10 300 python tools/symmetry_timing.py $n > $out/symmetry_$n.json 2> $out/symmetry_$n.err || { tail -3 $out/symmetry_$n.err; exit 1; }
  cat $out/symmetry_$n.json
done
for n in 160000 1048576; do
  CLOUDSC2_PLACE=0 PMC_NGPTOT=$n timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$n -- python3 tools/pmc_workload.py > $out/pmc_fetch_$n.log 2>&1 || { tail -5 $out/pmc_fetch_$n.log; exit 1; }
  CLOUDSC2_PLACE=0 PMC_NGPTOT=$n timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$n -- python3 tools/pmc_workload.py > $out/pmc_write_$n.log 2>&1 || { tail -5 $out/pmc_write_$n.log; exit 1; }
  python tools/pmc_parse.py $out/pmc_fetch_$n $out/pmc_write_$n $n > $out/pmc_traffic_$n.json && python -c "import json; d=json.load(open('$out/pmc_traffic_$n.json')); print($n, {k: round(v['traffic_over_algorithmic'], 4) for k, v in d['kernels'].items()})"
done
