#!/bin/bash
# address-translation counters of tl_kernel / nl_kernel at NPROMA 32 vs 128 (rocprofv3 --pmc, csv, four TCP counters per pass)
tag=${1:-r02_v}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp CLOUDSC2_PLACE=0
for k in tl nl; do for np in 32 128; do
  i=0
  for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
             "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
             "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    NPROMA=$np timeout -k 5 70 rocprofv3 --pmc $set --output-format csv -d $out/${k}_${np}_$i -- python3 tools/nl_workload.py 160000 $k 3 > $out/${k}_${np}_$i.log 2>&1 || { echo "failed $k $np set $i"; grep -m2 "F2026\|rror" $out/${k}_${np}_$i.log | cut -c1-200; }
  done
  echo "== $k NPROMA $np"; python3 tools/pmc_sq_parse.py ${k}_kernel $out/${k}_${np}_1 $out/${k}_${np}_2 $out/${k}_${np}_3
done; done
