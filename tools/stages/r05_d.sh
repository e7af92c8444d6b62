# A/B of the evaporation block's arithmetic (C2_EVAP_FAST=0|1), fresh processes, interleaved, 3 rounds
out=gpurun_out/r05_d; mkdir -p $out; : > $out/ab.txt
for r in 1 2 3; do for kn in "nl 160000" "nl 1048576" "tl 160000" "ad 160000"; do set -- $kn; for lib in evap_ieee evap_fast; do
  CLOUDSC2_LIB=$PWD/variants/$lib.so timeout -k 10 200 python bench.py --kernel $1 --ngptot $2 --levapls2 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/tmp.json 2> $out/tmp.err || { echo FAILED $lib $kn; tail -3 $out/tmp.err; exit 1; }
  python -c "import json; d=json.load(open('$out/tmp.json')); r=d['roofline']; print('$1 $2 $lib', round(r['kernel_ms_avg'],4), round(r['frac'],4))" | tee -a $out/ab.txt
done; done; done
# parity of the fast block against the reference (evaporation cases only) with the shipped library
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "evap or levapls2 or fuzz" > $out/pytest_evap.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_evap.log
