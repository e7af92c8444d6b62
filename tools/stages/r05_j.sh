# one gpurun call of round 5: the bench tests on the final bench.py, and the default line
out=gpurun_out/r05_j; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_gpu_bench.py -m gpu -q -x > $out/pytest_bench.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_bench.log
t0=$(date +%s); timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python -c "
import json; d=json.load(open('$out/bench_default.json')); print(json.dumps(d['baseline_configs_2_3'])[:900])"
