#!/bin/bash
# N fresh processes of the NL bench at 1 048 576 columns with the allocator's report on stderr (compositions' probe times, the chosen
# one probed again after the others are released) and the kernel's time; then tools/probe_vs_kernel.py at that size (5 unplaced
# hipMalloc arenas of 42.5 GB).  usage: tools/session_1m_verbose.sh TAG [N]
tag=${1:-v1m}; n=${2:-6}; out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  CLOUDSC2_PLACE_VERBOSE=1 timeout -k 10 300 python bench.py --kernel nl --ngptot 1048576 --steps 30 --warmup 3 --no-cpu-baseline --no-companions > $out/b.json 2>$out/err.log || { tail -3 $out/err.log; exit 1; }
  python -c "import json; d=json.load(open('$out/b.json')); r=d['roofline']; print('nl 1048576 kernel ms avg', round(r['kernel_ms_avg'],4), 'first/last tenth', round(r['kernel_ms_first_tenth'],4), round(r['kernel_ms_last_tenth'],4))"
  grep "cloudsc2_device_malloc" $out/err.log | cut -c1-600
done | tee $out/summary.txt
timeout -k 10 200 python tools/probe_vs_kernel.py 5 1048576 2>&1 | grep -v amdgpu.ids | tee $out/probe_vs_kernel_1m.txt
