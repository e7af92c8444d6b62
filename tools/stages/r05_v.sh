# one gpurun call of round 5: where the first host-array driver call's time goes
out=gpurun_out/r05_v; mkdir -p $out
timeout -k 10 300 python tools/debug/first_call.py > $out/first_call.log 2>&1; grep -v amdgpu $out/first_call.log
