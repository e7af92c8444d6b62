#!/bin/bash
# SQ occupancy/stall counters of the three kernels: tools/sq_profile.sh OUTDIR "sizes" "kernels"
# (separate rocprofv3 --pmc passes, <= 8 SQ counters each; no tracing domains)
out=${1:-gpurun_out/sq}; sizes=${2:-"160000 1048576"}; kernels=${3:-"nl tl ad"}
mkdir -p $out
for k in $kernels; do for n in $sizes; do
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
     --output-format csv -d $out/${k}_${n}_a -- python3 tools/nl_workload.py $n $k 3 > $out/${k}_${n}_a.log 2>&1 || exit 1
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU \
     --output-format csv -d $out/${k}_${n}_b -- python3 tools/nl_workload.py $n $k 3 > $out/${k}_${n}_b.log 2>&1 || exit 1
  echo "done $k $n"
done; done
