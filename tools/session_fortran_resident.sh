#!/bin/bash
# The Fortran mains on the device-resident state (CLOUDSC2_RESIDENT=1), fresh processes: what the boundary north_star names delivers on its
# first allocation.  usage: tools/session_fortran_resident.sh TAG [N]
tag=${1:-fr}; n=${2:-4}; out=$PWD/gpurun_out/$tag; mkdir -p $out; bld=$PWD/dwarf_p_cloudsc2_tl_ad_amd/fortran/build
cd $out
for i in $(seq 1 $n); do
  for args in "nl 1 160000 128" "nl 1 160000 32" "nl 1 1048576 128"; do
    set -- $args
    CLOUDSC2_RESIDENT=1 CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 $bld/dwarf-cloudsc2-$1 $2 $3 $4 > run.log 2> run.err || { echo "FAILED $args"; tail -3 run.err; exit 1; }
    echo "dwarf-cloudsc2-$1 $2 $3 $4 (resident):"; grep -i -A3 "NUMPROC=\|Time(usec)\|columns/s\|MFlops" run.err run.log | grep -v "^--" | head -8 | cut -c1-200
  done
done | tee summary.txt
