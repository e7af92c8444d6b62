#!/bin/bash
# BASELINE.json configs[1]: NGPTOT=160000, NPROMA sweep 32-256, all three kernels; plus the evaporation branch on.
# usage: tools/nproma_sweep.sh OUT.json
out=${1:-gpurun_out/nproma_sweep.json}
echo "[" > $out; first=1
for k in nl tl ad; do for np in 32 64 128 256; do for ev in "" "--levapls2"; do
  [ -n "$ev" ] && [ $np != 128 ] && continue
  line=$(timeout -k 10 200 python bench.py --kernel $k --nproma $np $ev --steps 30 --warmup 3 --no-cpu-baseline --no-companions 2>/dev/null) || exit 1
  [ $first = 1 ] || echo "," >> $out; first=0
  echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r={'kernel':'$k','nproma':$np,'levapls2':bool('$ev'),'ngptot':d['config']['ngptot_per_gpu'],'ms':d['roofline']['kernel_ms_avg'],'columns_per_s':d['value'],'frac_of_8TBps':d['roofline']['frac']}; print(json.dumps(r)); print('$k nproma=$np $ev', round(r['ms'],3), round(r['frac_of_8TBps'],3), file=sys.stderr)" >> $out
done; done; done
echo "]" >> $out
