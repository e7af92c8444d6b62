# one gpurun call of round 5: the bench line measuring its own HBM traffic
out=gpurun_out/r05_n; mkdir -p $out
t0=$(date +%s); timeout -k 10 600 python bench.py --steps 50 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']
print(r.get('traffic_measurement_failed')); print(r['traffic_source'][:200]); print(r['traffic_over_algorithmic'], r['frac_actual_bytes'], r.get('traffic_committed_pass'))
for k,v in d['companion_kernels'].items(): print(k, v['traffic_over_algorithmic'], v['frac_actual_bytes'], v['traffic_source'][:40])"
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_multirank.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
