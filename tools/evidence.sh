#!/bin/bash
# One GPU session that regenerates the judged evidence: tests, smoke, default bench line, rocprofv3 kernel stats of the same bench
# command at 160 000 and at 1 048 576 columns, PMC traffic passes (separate --pmc runs), TL/AD and 1M-column bench lines.
# usage: tools/evidence.sh TAG [a|b|all]   (outputs under gpurun_out/TAG/; tools/collect_profiles.py copies the summaries to profiles/)
# stage a: tests, smoke, the default line, rocprofv3 kernel-trace of every timed kernel; stage b: the PMC passes and the other bench
# lines (one gpurun call holds at most 20 minutes: a and b are two calls)
tag=${1:-ev}; stage=${2:-all}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "$stage" != b ]; then
du -a --max-depth=3 . 2>/dev/null | sort -n | tail -60 > $out/du_on_the_box.txt   # what travelled (.gpurunignore)
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 && echo smoke ok || { echo smoke FAILED; tail -5 $out/smoke.log; }
# what the box is and how it clocks while the kernel runs (boxes of the pool measure up to 12 % apart)
rocm-smi --showproductname --showclocks --showpower --showperflevel --showmemvendor > $out/rocm_smi_idle.txt 2>&1
(for i in 1 2 3 4 5 6 7 8; do sleep 2; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature" | tr '\n' ';'; echo; done > $out/rocm_smi_during_bench.txt) &
smi_pid=$!
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err && cut -c1-600 $out/bench_default.json || exit 1
wait $smi_pid
# rocprofv3 --stats of the same bench command (no child processes under the profiler); the library's placement probes show up as
# place_probe_kernel, the hot kernel's average must agree with the HIP-event figure of the JSON line
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline --no-companions > $out/bench_prof.json 2> $out/bench_prof.err || exit 1
python tools/trace_timed_region.py $out/prof/*/*_kernel_trace.csv nl_kernel 1000 > $out/bench_prof_timed_region.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_1m -- python3 bench.py --ngptot 1048576 --steps 50 --no-cpu-baseline --no-companions > $out/bench_prof_1m.json 2> $out/bench_prof_1m.err || exit 1
python tools/trace_timed_region.py $out/prof_1m/*/*_kernel_trace.csv nl_kernel 50 > $out/bench_prof_1m_timed_region.json
# rocprofv3 kernel-trace of EVERY timed kernel, not NL alone: the same bench command per kernel, the program
# directly after `--`, no child processes; the last K dispatches are the timed region (tools/kernel_stats_rows.py)
specs=""
for n in 160000 1048576; do
  for kf in "tl:--kernel tl:tl_kernel:57072" "ad:--kernel ad:ad_kernel:85608" "adrev:--kernel ad --ad-sweep reverse --ad-assign:ad_reverse_kernel:59264" "nlevap:--kernel nl --levapls2:nl_kernel:28536"; do
    lab=${kf%%:*}; rest=${kf#*:}; flags=${rest%%:*}; rest=${rest#*:}; pat=${rest%%:*}; bpc=${rest#*:}
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${lab}_$n -- python3 bench.py $flags --ngptot $n --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/bench_prof_${lab}_$n.json 2> $out/bench_prof_${lab}_$n.err || exit 1
    specs="$specs ${lab}_$n:$(ls $out/prof_${lab}_$n/*/*_kernel_trace.csv):$pat:30:$n:$bpc"
  done
  # the two self-tests as whole driver calls on a resident state: taylor_kernel (the lambda sweep), tl_kernel with C2F_TRAJ|C2F_SELFINC,
  # ad_reverse_kernel with C2F_ADNORM; each test runs 4 times, the last 3 are reported
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_selftests_$n -- python3 bench.py --self-tests --ngptot $n > $out/bench_prof_selftests_$n.json 2> $out/bench_prof_selftests_$n.err || exit 1
  specs="$specs taylor_$n:$(ls $out/prof_selftests_$n/*/*_kernel_trace.csv):taylor_kernel:3:$n:0"
done
specs="nl_160000:$(ls $out/prof/*/*_kernel_trace.csv):nl_kernel:1000:160000:28536 nl_1048576:$(ls $out/prof_1m/*/*_kernel_trace.csv):nl_kernel:50:1048576:28536 $specs"
python tools/kernel_stats_rows.py $out/timed_kernel_stats.csv $specs
fi
[ "$stage" = a ] && exit 0
for n in 160000 1048576; do
  CLOUDSC2_PLACE=0 PMC_NGPTOT=$n timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$n -- python3 tools/pmc_workload.py > $out/pmc_fetch_$n.log 2>&1 || exit 1
  CLOUDSC2_PLACE=0 PMC_NGPTOT=$n timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$n -- python3 tools/pmc_workload.py > $out/pmc_write_$n.log 2>&1 || exit 1
  python tools/pmc_parse.py $out/pmc_fetch_$n $out/pmc_write_$n $n > $out/pmc_traffic_$n.json
done
for k in nl tl ad; do for n in 160000 1048576; do
  timeout -k 10 300 python bench.py --kernel $k --ngptot $n --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/bench_${k}_$n.json 2>/dev/null
  python -c "import json; d=json.load(open('$out/bench_${k}_$n.json')); print('$k $n', round(d['ms_per_step'],3), '%.3e'%d['value'], round(d['roofline']['frac'],3))"
done; done
# the adjoint's other forms: assign (x = A^T y), the reverse sweep alone, and both (the AD leg of the adjoint test)
for n in 16384 160000 1048576; do for form in "adassign:--ad-assign" "adreverse:--ad-sweep reverse" "adreverseassign:--ad-sweep reverse --ad-assign"; do
  timeout -k 10 300 python bench.py --kernel ad ${form#*:} --ngptot $n --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/bench_${form%%:*}_$n.json 2>/dev/null
  python -c "import json; d=json.load(open('$out/bench_${form%%:*}_$n.json')); r=d['roofline']; print('${form%%:*} $n', round(r['kernel_ms_avg'],3), r['bytes_per_column'], round(r['frac'],3))"
done; done
for n in 16384 160000 1048576; do timeout -k 10 300 python tools/symmetry_timing.py $n > $out/symmetry_$n.json 2>/dev/null && cat $out/symmetry_$n.json; done
# the whole Taylor test (SATUR, NL, TL, the lambda sweep, block sums) in precise and in fast arithmetic
for m in 2 1; do timeout -k 10 300 python tools/taylor_ab.py $m >> $out/taylor_test.jsonl 2>/dev/null; done; cat $out/taylor_test.jsonl
