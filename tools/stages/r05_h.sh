# does the progress priority (nl_fair) pay for the evaporation NL variant, which runs two waves per SIMD (its launch is NOT one round)?
out=gpurun_out/r05_h; mkdir -p $out; : > $out/ab.txt
python - <<'PY' | tee -a $out/ab.txt
import ctypes as C, sys
sys.path.insert(0,'.')
from dwarf_p_cloudsc2_tl_ad_amd import binding as B
v=C.c_int()
for name,k,f in (("nl plain off32",0,32),("nl evap off32",0,36),("nl evap 64-bit offsets",0,4),("nl plain 64-bit",0,0)):
    B.check(B.lib.cloudsc2_kernel_occupancy(k,f,C.byref(v))); print("occupancy", name, v.value, "workgroups per CU")
PY
for r in 1 2 3; do for n in 160000 100000 65536; do for fair in default 0; do
  if [ $fair = default ]; then unset CLOUDSC2_FAIR; else export CLOUDSC2_FAIR=0; fi
  timeout -k 10 200 python bench.py --kernel nl --ngptot $n --levapls2 --steps 100 --warmup 3 --no-cpu-baseline --no-companions > $out/tmp.json 2> $out/tmp.err || { echo FAILED; tail -3 $out/tmp.err; exit 1; }
  python -c "import json; d=json.load(open('$out/tmp.json')); r=d['roofline']; print('nl-evap $n fair=$fair', round(r['kernel_ms_avg'],4), round(r['frac'],4))" | tee -a $out/ab.txt
done; done; done
