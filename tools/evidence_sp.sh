#!/bin/bash
# fp32 (CLOUDSC2_PRECISION=single) counterpart of tools/evidence.sh: bench lines, rocprofv3 kernel stats of the bench
# command, PMC traffic passes.  usage: tools/evidence_sp.sh TAG   (outputs under gpurun_out/TAG/)
tag=${1:-evsp}; out=gpurun_out/$tag; mkdir -p $out
export CLOUDSC2_PRECISION=single
timeout -k 10 300 python tests/single_checks.py gpu > $out/single_checks_gpu.log 2>&1 && tail -1 $out/single_checks_gpu.log || { tail -5 $out/single_checks_gpu.log; exit 1; }
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err && cat $out/bench_default.json || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-companions > $out/bench_prof.json 2> $out/bench_prof.err || exit 1
python tools/trace_timed_region.py $out/prof/*/*_kernel_trace.csv nl_kernel 1000 > $out/bench_prof_timed_region.json
for n in 160000 1048576; do
  CLOUDSC2_PLACE=0 PMC_NGPTOT=$n timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$n -- python3 tools/pmc_workload.py > $out/pmc_fetch_$n.log 2>&1 || exit 1
  CLOUDSC2_PLACE=0 PMC_NGPTOT=$n timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$n -- python3 tools/pmc_workload.py > $out/pmc_write_$n.log 2>&1 || exit 1
  python tools/pmc_parse.py $out/pmc_fetch_$n $out/pmc_write_$n $n 4 > $out/pmc_traffic_$n.json
done
for k in nl tl ad; do for n in 160000 1048576; do
  timeout -k 10 300 python bench.py --kernel $k --ngptot $n --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/bench_${k}_$n.json 2>/dev/null
  python -c "import json; d=json.load(open('$out/bench_${k}_$n.json')); print('$k $n', round(d['ms_per_step'],3), '%.3e'%d['value'], round(d['roofline']['frac'],3))"
done; done
