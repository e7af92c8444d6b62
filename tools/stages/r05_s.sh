# one gpurun call of round 5: the one case of the second fuzz seed that passes 1e-11 by 1 % (nlev 200), field by field, in both arithmetic modes
out=gpurun_out/r05_s; mkdir -p $out
timeout -k 10 200 python tools/debug/fuzz_tlad_case.py 7 2605 > $out/case7_fast.log 2>&1; cat $out/case7_fast.log | grep -v amdgpu
CLOUDSC2_MATH=precise timeout -k 10 200 python tools/debug/fuzz_tlad_case.py 7 2605 > $out/case7_precise.log 2>&1; grep "^TL\|^AD" $out/case7_precise.log | sort -k4 -g -r | head -6
