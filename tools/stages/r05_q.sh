# one gpurun call of round 5: the tests touched after the final suite run
out=gpurun_out/r05_q; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -m gpu -q -x -k "bench or default_line or capture or two_ranks or host_array" > $out/pytest.log 2>&1; echo "rc=$?"; tail -3 $out/pytest.log
