#!/bin/bash
# GPU session: characterise the HBM-placement effect.  usage: tools/session_placement.sh TAG
tag=${1:-r02_a}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocm-smi --showmemvendor --showproductname --showvbios > $out/rocm_smi.txt 2>&1
timeout -k 10 240 tools/bin/hbm_probe all 150 > $out/probe_all.txt 2>&1 || { echo probe failed; tail -5 $out/probe_all.txt; exit 1; }
tail -45 $out/probe_all.txt
timeout -k 10 400 python tools/placement_layout.py 14 > $out/layout.txt 2>&1 || { echo layout failed; tail -5 $out/layout.txt; exit 1; }
cat $out/layout.txt
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_BUSY TCC_TAG_STALL" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_IO_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_64B TCC_EA0_WR_UNCACHED_32B TCC_EA0_WRREQ_DRAM" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT GRBM_EA_BUSY"; do
  name=$(echo $set | tr ' ' '+' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format json csv -d $out/pmc_$name -- tools/bin/hbm_probe pmc 30 > $out/pmc_$name.log 2>&1 || { echo "pmc $set failed"; tail -3 $out/pmc_$name.log; }
done
ls -R $out | head -40
