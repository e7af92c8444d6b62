#!/bin/bash
# why is TL (and AD) 40-80 % slower at NPROMA=32 when NL is not?  memory-side counters of tl_kernel at NPROMA 32 vs 128
# (rocprofv3 --pmc, csv, at most four counters of one block per pass; CLOUDSC2_PLACE=0: no placement probes under the profiler)
tag=${1:-r02_u}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp CLOUDSC2_PLACE=0
for np in 32 128; do
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr"; do
    i=$((i+1))
    NPROMA=$np timeout -k 5 70 rocprofv3 --pmc $set --output-format csv -d $out/tl_${np}_$i -- python3 tools/nl_workload.py 160000 tl 3 > $out/tl_${np}_$i.log 2>&1 || { echo "failed tl $np set $i"; grep -m2 "F2026\|rror" $out/tl_${np}_$i.log | cut -c1-200; }
  done
  echo "== tl NPROMA $np"; python3 tools/pmc_sq_parse.py tl_kernel $out/tl_${np}_1 $out/tl_${np}_2 $out/tl_${np}_3 $out/tl_${np}_4
done
