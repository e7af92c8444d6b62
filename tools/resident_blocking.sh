#!/bin/bash
# GPU suite; the Fortran mains on the resident state at the caller's NPROMA 32 / 100 / 128 (device arrays blocked
# by the library), fresh process per run; with CLOUDSC2_STATE_NPROMA=0 (device arrays in the caller's blocking) for comparison.
tag=${1:-resident}; out=$PWD/gpurun_out/$tag; mkdir -p $out; bld=$PWD/dwarf_p_cloudsc2_tl_ad_amd/fortran/build
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=6 > $out/pytest_gpu.log 2>&1; tail -12 $out/pytest_gpu.log
cd $out
for rep in 1 2; do
for own in "" 0; do
  for k in nl tl ad; do
    for np in 32 100 128; do
      if [ -z "$own" ]; then unset CLOUDSC2_STATE_NPROMA; else export CLOUDSC2_STATE_NPROMA=$own; fi
      CLOUDSC2_RESIDENT=1 timeout -k 10 300 $bld/dwarf-cloudsc2-$k 1 160000 $np > run.log 2> run.err || { echo "FAILED $k $np"; tail -5 run.err; exit 1; }
      ms=$(grep "GPU kernel" run.err | head -1 | awk '{print $3}')
      verdict=$(grep -h -i "TEST PASSED\|TEST OK\|TEST FAILED" run.log run.err | head -1 | cut -c1-60)
      echo "dwarf-cloudsc2-$k 1 160000 $np resident, device blocking ${own:-default}: $ms ms  $verdict"
    done
  done
done
done | tee fortran_resident_blocking.txt
grep -A5 "NUMOMP" run.err | head -8
