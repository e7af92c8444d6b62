# one gpurun call of round 5: why is the fp32 TL sweep at 1 M columns unpaced (5.79 ms against round 4's 5.02)?
out=gpurun_out/r05_k; mkdir -p $out
export CLOUDSC2_PRECISION=single
python - <<'PY' 2>&1 | tee $out/probe.txt
import ctypes as C, sys
sys.path.insert(0,'.')
import torch
from dwarf_p_cloudsc2_tl_ad_amd import binding as B
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
v=C.c_int()
for name,k,f in (("tl traj off32",1,1|8|32),("tl traj 64-bit",1,1|8),("ad off32",2,1|32),("ad 64-bit",2,1),("adrev off32",3,1|32),("adrev 64-bit",3,1)):
    rc=B.lib.cloudsc2_kernel_occupancy(k,f,C.byref(v)); print("occupancy", name, v.value if rc==0 else ("rc",rc))
for pc in (1,2,3,4,5,6):
    for _ in range(2):
        a,b=C.c_longlong(),C.c_longlong()
        rc=B.lib.cloudsc2_pace_probe(pc,C.byref(a),C.byref(b))
        print("pace probe per_cu",pc,"rc",rc,"checked",a.value,"wrong",b.value, (B.lib.cloudsc2_last_error() or b"").decode() if rc else "")
PY
CLOUDSC2_PACE_VERBOSE=1 timeout -k 10 200 python bench.py --precision single --kernel tl --ngptot 1048576 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/tl_1m.json 2> $out/tl_1m.err; grep -c paced $out/tl_1m.err; grep "cloudsc2:" $out/tl_1m.err | sort | uniq -c | head
python -c "import json; d=json.load(open('$out/tl_1m.json')); print(d['roofline']['kernel_ms_avg'])"
CLOUDSC2_PACE=0 timeout -k 10 200 python bench.py --precision single --kernel tl --ngptot 1048576 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/tl_1m_nopace.json 2>/dev/null; python -c "import json; d=json.load(open('$out/tl_1m_nopace.json')); print('PACE=0', d['roofline']['kernel_ms_avg'])"
