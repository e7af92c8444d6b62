# one gpurun call of round 5: A/B of the tangent / adjoint evaporation blocks (C2_EVAP_FAST_TLAD=0|1), fresh processes, interleaved, 3 rounds; parity of the shipped library
out=gpurun_out/r05_l; mkdir -p $out; : > $out/ab.txt
for r in 1 2 3; do for kn in "tl 160000" "ad 160000" "tl 1048576" "ad 1048576"; do set -- $kn; for lib in evap_tlad_ieee evap_tlad_fast; do
  CLOUDSC2_LIB=$PWD/variants/$lib.so timeout -k 10 200 python bench.py --kernel $1 --ngptot $2 --levapls2 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/tmp.json 2> $out/tmp.err || { echo FAILED $lib $kn; tail -3 $out/tmp.err; exit 1; }
  python -c "import json; d=json.load(open('$out/tmp.json')); r=d['roofline']; print('$1 $2 $lib', round(r['kernel_ms_avg'],4), round(r['frac'],4))" | tee -a $out/ab.txt
done; done; done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_single.py -m gpu -q -x -k "evap or levapls2 or fuzz or single or two_sweeps or both_sequences or lambda_sweep or increments_formed or norms_formed" > $out/pytest_evap.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_evap.log
