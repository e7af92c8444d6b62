# one gpurun call of round 5: experiment -- the partial-round pacing for the NL sweep's multi-round launches (CLOUDSC2_NL_PACE=1), A/B, three interleaved rounds
out=gpurun_out/r05_u; mkdir -p $out; : > $out/ab.txt
CLOUDSC2_NL_PACE=1 CLOUDSC2_PACE_VERBOSE=1 timeout -k 10 200 python bench.py --ngptot 1048576 --steps 5 --warmup 1 --no-cpu-baseline --no-companions 2>&1 >/dev/null | grep "cloudsc2:" | sort | uniq -c | head -5 | tee -a $out/ab.txt
for r in 1 2 3; do for n in 1048576 524288 400000 300000 230000; do for p in 0 1; do
  CLOUDSC2_NL_PACE=$p timeout -k 10 200 python bench.py --ngptot $n --steps 50 --warmup 3 --no-cpu-baseline --no-companions > $out/tmp.json 2> $out/tmp.err || { echo FAILED; tail -3 $out/tmp.err; exit 1; }
  python -c "import json; d=json.load(open('$out/tmp.json')); r=d['roofline']; print('nl $n NL_PACE=$p', round(r['kernel_ms_avg'],4), round(r['frac'],4))" | tee -a $out/ab.txt
done; done; done
