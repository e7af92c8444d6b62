#!/bin/bash
# GPU session: where are the places of the HBM that take writes 10-20 % slower, and what do the L2 -> fabric counters say?
tag=${1:-r02_b}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocm-smi --showmemorypartition --showcomputepartition > $out/partition.txt 2>&1
timeout -k 10 300 tools/bin/hbm_map map 256 512 > $out/map_512.txt 2>&1 || { echo map failed; tail -5 $out/map_512.txt; exit 1; }
cat $out/map_512.txt
i=0
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_BUSY TCC_TAG_STALL" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_IO_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_64B TCC_EA0_WR_UNCACHED_32B TCC_EA0_WRREQ_DRAM" \
           "TCC_EA0_WRREQ_WRITE_DRAM TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_GMI_32B TCC_EA0_WRREQ_WRITE_IO_32B"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 200 rocprofv3 --pmc $set --output-format json -d /tmp/pmc_$i -- $OLDPWD/tools/bin/hbm_map pmc 48 512 > $OLDPWD/$out/pmc_$i.log 2>&1 ) || { echo "pmc set $i failed"; tail -3 $out/pmc_$i.log; continue; }
  f=$(ls /tmp/pmc_$i/*/*_results.json | head -1)
  python3 tools/pmc_json_summary.py $f 'fill_tag<[12]>' > $out/pmc_summary_$i.json 2> $out/pmc_summary_$i.err || { echo "summary $i failed"; tail -3 $out/pmc_summary_$i.err; }
  python3 - $out/pmc_summary_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.get("kernels", {}).items():
    for c, s in v["counters"].items():
        print(f"{k[:24]:24s} {c:38s} n={s['instances']:4d} sum={s['sum']:14.0f} min={s['min']:10.0f} max={s['max']:10.0f}")
PY
done
du -sh $out
