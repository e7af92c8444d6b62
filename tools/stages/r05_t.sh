# one gpurun call of round 5: the launchers from four host threads at once
out=gpurun_out/r05_t; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "four_host_threads" > $out/pytest.log 2>&1; echo "rc=$?"; tail -15 $out/pytest.log
