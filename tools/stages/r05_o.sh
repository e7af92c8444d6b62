# one gpurun call of round 5: where the GPU suite's time goes
out=gpurun_out/r05_o; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=25 > $out/pytest_durations.log 2>&1; echo "rc=$?"; grep -A30 "slowest" $out/pytest_durations.log | head -34; tail -2 $out/pytest_durations.log
