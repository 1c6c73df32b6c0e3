#!/bin/bash
# Rehearsal of both ways the driver may start a several-GPU bench, with the ranks sharing the one GPU over gloo
# (timings mean nothing; what is checked: the flow runs to the end, one JSON line, exact results).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/rehearse; mkdir -p $O
export CGPS_BENCH_REHEARSAL_GLOO=1 CGPS_BENCH_PREWARM_STEPS=20
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 > $O/torchrun_2.json 2> $O/torchrun_2.err; echo "torchrun 2 ranks rc=$?"
timeout -k 10 400 python3 bench.py --gpus 4 --steps 10 --warmup 3 > $O/self_4.json 2> $O/self_4.err; echo "self-launch 4 ranks rc=$?"
python - <<'PY'
import json
for f in ('torchrun_2','self_4'):
    try:
        d=json.load(open('gpurun_out/rehearse/%s.json'%f))
        print(f, 'n_gpus', d['n_gpus'], d['scaling'], 'rows/gpu', d['config']['rows_per_gpu'], 'check', d['check'], 'cpu_baseline' in d and d['cpu_baseline'] is not None, 'traffic', d['roofline'].get('traffic'), list(d['extras'].keys()))
    except Exception as e: print(f, 'ERR', e)
PY
tail -n 3 $O/torchrun_2.err; tail -n 3 $O/self_4.err
