# one gpurun call of round 5 (kept as the record of what produced gpurun_out/r05_a and the profiles/r05_* files derived from it)
out=gpurun_out/r05_a; mkdir -p $out
CLOUDSC2_PACE_VERBOSE=1 timeout -k 10 120 python tools/capture_probe.py cold > $out/capture_cold.log 2>&1; echo "cold rc=$?" 
CLOUDSC2_PACE_VERBOSE=1 timeout -k 10 120 python tools/capture_probe.py warm > $out/capture_warm.log 2>&1; echo "warm rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pace or dispatch or capture or pacing" > $out/pytest_probe.log 2>&1; tail -5 $out/pytest_probe.log
for k in tl ad; do CLOUDSC2_PACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel $k --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/bench_$k.json 2> $out/bench_$k.err; python -c "import json; d=json.load(open('$out/bench_$k.json')); print('$k', d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"; grep -c "paced" $out/bench_$k.err; done
