#!/bin/bash
# The Taylor test through the Fortran main on a resident state: kernel time of the whole driver call (SATUR, NL, increments, TL,
# the lambda sweep, block sums) at the caller's NPROMA 32 / 100 / 128 and at the reference's own size (100 columns, NPROMA 1 / 32).
#   tools/taylor_timing.sh TAG ["pytest -k expression"]
tag=${1:-taylor}; out=$PWD/gpurun_out/$tag; mkdir -p $out; bld=$PWD/dwarf_p_cloudsc2_tl_ad_amd/fortran/build
export TMPDIR=/tmp
if [ -n "$2" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$2" > $out/pytest_gpu.log 2>&1; rc=$?; tail -15 $out/pytest_gpu.log
  [ $rc = 0 ] || exit $rc
fi
cd $out
for rep in 1 2; do
  for cfg in "160000 32" "160000 100" "160000 128" "100 1" "100 32" "1048576 128"; do
    set -- $cfg
    CLOUDSC2_RESIDENT=1 timeout -k 10 300 $bld/dwarf-cloudsc2-tl 1 $1 $2 > run.log 2> run.err || { echo "FAILED $cfg"; tail -5 run.err; exit 1; }
    ms=$(grep "GPU kernel" run.err | head -1 | awk '{print $3}')
    verdict=$(grep -h -i "TEST PASSED\|TEST FAILED" run.log run.err | head -1 | cut -c1-60)
    echo "dwarf-cloudsc2-tl 1 $1 $2 resident: $ms ms  $verdict"
  done
done | tee taylor_timing.txt
