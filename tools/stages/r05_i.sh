# one gpurun call of round 5 (kept as the record of what produced gpurun_out/r05_i and the profiles/r05_* files derived from it)
out=gpurun_out/r05_i; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "target_size or full_size_tl_ad" > $out/pytest_1m.log 2>&1; echo "rc=$?"; grep -E "columns|passed|failed|Error" $out/pytest_1m.log | tail -12
