#!/bin/bash
# Second evidence session: NPROMA sweep on the final kernels (BASELINE configs[1]), SQ counters of the three kernels at both sizes,
# and a 2-rank rehearsal of `bench.py --gpus 2` on the one GPU of the box (gloo backend: the ranks share the device).
# usage: tools/evidence2.sh TAG
tag=${1:-ev2}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
tools/nproma_sweep.sh $out/nproma_sweep.json 2> $out/nproma_sweep.txt; cat $out/nproma_sweep.txt
CLOUDSC2_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 50 --warmup 3 > $out/bench_2ranks_gloo.json 2> $out/bench_2ranks_gloo.err; cut -c1-300 $out/bench_2ranks_gloo.json; tail -2 $out/bench_2ranks_gloo.err
CLOUDSC2_PLACE=0 tools/sq_profile.sh $out/sq "160000 1048576" "nl tl ad" > $out/sq.log 2>&1
for k in nl tl ad; do for n in 160000 1048576; do echo "== ${k}_kernel $n columns"; python tools/pmc_sq_parse.py ${k}_kernel $out/sq/${k}_${n}_a $out/sq/${k}_${n}_b; done; done > $out/sq_counters.txt
tail -20 $out/sq_counters.txt
