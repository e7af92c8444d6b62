#!/bin/bash
# Round 3, session b: GPU suite (with the fuzz slice and the reference-pinned tiler / validator tests), then the NL instruction-count
# candidates A/B (fresh process per run, interleaved, medians): csrc/variants/*.so built by hand with the macros named in the files.
tag=${1:-r03_b}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=8 > $out/pytest_gpu.log 2>&1; tail -14 $out/pytest_gpu.log
V=dwarf_p_cloudsc2_tl_ad_amd/csrc/variants
timeout -k 10 1500 python tools/ab_kernels.py "$V/base.so $V/rh.so $V/rh_sat.so $V/rh_defer8.so" "nl" "160000 1048576" 5 > $out/ab_nl.txt 2>&1; cat $out/ab_nl.txt
