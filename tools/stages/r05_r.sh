# one gpurun call of round 5: the long randomised parity sweep on the round's final kernels (two seeds)
out=gpurun_out/r05_r; mkdir -p $out
timeout -k 10 900 python tests/fuzz_parity.py 160 777 > $out/fuzz_777.log 2>&1; echo "rc=$?"; tail -1 $out/fuzz_777.log
timeout -k 10 900 python tests/fuzz_parity.py 160 2605 > $out/fuzz_2605.log 2>&1; echo "rc=$?"; tail -1 $out/fuzz_2605.log
grep -c "levapls2.: True\|ldrain1d.: True" $out/fuzz_777.log $out/fuzz_2605.log
