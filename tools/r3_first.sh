#!/bin/bash
# Round-3 first GPU pass: full -m gpu suite, the driver's bench command, the two-rank gloo rehearsal of
# `bench.py --gpus 2` (self-launching), the per-rank shard figure.  Output under gpurun_out/r03a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
CGPS_BENCH_REHEARSAL_GLOO=1 CGPS_BENCH_PREWARM_STEPS=50 timeout -k 10 300 python3 bench.py --gpus 2 --rows 4194304 --steps 20 --warmup 5 > $O/rehearsal_gloo_2ranks.json 2> $O/rehearsal.err; echo "rehearsal rc=$?"
CGPS_BENCH_FORCE_SHARDED=1 timeout -k 10 200 python bench.py --rows $((1<<21)) --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_2p21.json 2> $O/shard_2p21.err; echo "shard rc=$?"
CGPS_BENCH_FORCE_SHARDED=1 timeout -k 10 200 python bench.py --rows $((1<<21)) --steps 200 --warmup 20 --no-cpu-baseline --sub-shards 2 > $O/shard_2p21_s2.json 2> $O/shard_2p21_s2.err; echo "shard s2 rc=$?"
CGPS_BENCH_FORCE_SHARDED=1 timeout -k 10 200 python bench.py --rows $((1<<21)) --steps 200 --warmup 20 --no-cpu-baseline --sub-shards 4 > $O/shard_2p21_s4.json 2> $O/shard_2p21_s4.err; echo "shard s4 rc=$?"
python - <<'PY'
import json
for f in ('bench_driver','rehearsal_gloo_2ranks','shard_2p21','shard_2p21_s2','shard_2p21_s4'):
    try:
        d=json.load(open('gpurun_out/r03a/%s.json'%f)); print(f, '%.2f us'%(d['ms_per_step']*1e3), 'frac', d.get('roofline_frac_whole_op'), d.get('extras',{}).get('headline_cold',{}).get('ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
