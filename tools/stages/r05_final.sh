# one gpurun call of round 5: the last confirmation on the committed sources -- the whole GPU suite, smoke, the driver's command
out=gpurun_out/r05_final; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 && echo smoke ok || { echo smoke FAILED; tail -5 $out/smoke.log; }
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-300 $out/bench_default.json
