# rehearsal of the N > 1 bench path with four ranks sharing the one GPU (gloo collectives; RCCL refuses ranks that share a device)
out=gpurun_out/r05_g; mkdir -p $out
t0=$(date +%s)
CLOUDSC2_DIST_BACKEND=gloo CLOUDSC2_BENCH_LOGDIR=$out/ranks timeout -k 10 500 python bench.py --gpus 4 --steps 200 --warmup 5 > $out/bench_4ranks.json 2> $out/bench_4ranks.err; echo "rc=$? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_g/bench_4ranks.json').read().strip().splitlines()[-1])
print(d['stage'], d['n_gpus'], d['value'], d['seconds_since_start'])
print({k:(v.get('value'), v.get('kernel_ms_avg_per_rank')) for k,v in d['companion_kernels'].items() if isinstance(v,dict)})
print(d.get('cpu_baseline',{}).get('value'), d['verdicts'].get('tl_passed'), d['verdicts'].get('ad_ok'), d['verdicts'].get('native_comm'))
PY
