# one gpurun call of round 5 (kept as the record of what produced gpurun_out/r05_e and the profiles/r05_* files derived from it)
out=gpurun_out/r05_e; mkdir -p $out
timeout -k 10 200 python tools/debug/evap_diff.py > $out/diff_fast.log 2>&1; tail -6 $out/diff_fast.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_single.py -m gpu -q -x -k "evap or levapls2 or fuzz or single or two_sweeps or both_sequences" > $out/pytest_evap.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_evap.log
for k in nl tl ad; do timeout -k 10 200 python bench.py --kernel $k --levapls2 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/bench_evap_$k.json 2>/dev/null; python -c "import json; d=json.load(open('$out/bench_evap_$k.json')); r=d['roofline']; print('$k evap', round(r['kernel_ms_avg'],4), round(r['frac'],4))"; done
