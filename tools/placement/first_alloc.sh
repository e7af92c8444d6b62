#!/bin/bash
# N fresh processes with and without the allocator's placement, alternating: tools/first_alloc.sh OUT [N [KERNEL [NGPTOT]]]
out=${1:-gpurun_out/first_alloc.txt}; n=${2:-6}; k=${3:-nl}; g=${4:-160000}
: > $out
for i in $(seq 1 $n); do
  CLOUDSC2_PLACE=1 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 >> $out || exit 1
  CLOUDSC2_PLACE=0 timeout -k 10 120 python tools/first_alloc.py $k $g 2>/dev/null | tail -1 >> $out || exit 1
done
cat $out
