# one gpurun call of round 5: the driver's own N > 1 command line (torch.distributed.run) rehearsed with two ranks sharing the one GPU (gloo)
out=gpurun_out/r05_p; mkdir -p $out
t0=$(date +%s)
CLOUDSC2_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 > $out/torchrun.out 2> $out/torchrun.err; echo "rc=$? in $(( $(date +%s) - t0 )) s"
grep -c '^{' $out/torchrun.out
python - <<'PY'
import json
lines=[json.loads(l) for l in open('gpurun_out/r05_p/torchrun.out') if l.startswith('{')]
for d in lines:
    print(d['stage'][:20], d['n_gpus'], round(d['value']), d.get('seconds_since_start'), sorted(k for k in d if k not in ('config','roofline')))
d=lines[-1]
print({k:(round(v['value']), v['kernel_ms_avg_per_rank']) for k,v in d['companion_kernels'].items() if isinstance(v,dict)}, d['cpu_baseline'].get('value'), d['verdicts']['tl_passed'], d['verdicts']['ad_ok'])
PY
tail -3 $out/torchrun.err
