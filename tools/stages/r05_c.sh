# one gpurun call of round 5 (kept as the record of what produced gpurun_out/r05_c and the profiles/r05_* files derived from it)
out=gpurun_out/r05_c; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_gpu_bench.py -m gpu -q -x > $out/pytest_bench.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_bench.log
t0=$(date +%s); timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"; cut -c1-400 $out/bench_default.json
