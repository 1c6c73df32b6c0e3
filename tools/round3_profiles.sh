#!/bin/bash
# Round-3 measurement set on the GPU box (writes under gpurun_out/r03p/ and profiles/r03_* on the box; the caller copies
# gpurun_out/r03p back): bench lines, rocprofv3 kernel table of the driver's command, PMC traffic of the headline and of
# the shard sizes of config 4 on 8 / 4 / 2 GPUs, the per-rank shard figure, config 4 on one GPU, the two-rank rehearsal.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03p; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py > $O/bench_default.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt3 && rocprofv3 --kernel-trace --stats -d /tmp/kt3 -o p --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $R/$O/headline_prof.log 2>&1; cp $(find /tmp/kt3 -name "*kernel_stats.csv" | head -1) $R/$O/kernel_stats_headline.csv )
bash tools/pmc_traffic.sh $O/traffic_headline
python tools/pmc_traffic_json.py $O/traffic_headline 3 > $O/pmc_traffic_json.log 2>&1
for n in $((1<<21)) $((1<<22)) $((1<<23)); do
  mkdir -p $O/traffic_shards/rows_$n
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && CGPS_BENCH_FORCE_SHARDED=1 CGPS_BENCH_PREWARM_STEPS=0 rocprofv3 --pmc $c --kernel-trace -d $R/$O/traffic_shards/rows_$n/$c -o p --output-format csv -- python3 $R/bench.py --rows $n --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/traffic_shards/rows_${n}_$c.log 2>&1 )
  done
done
python tools/pmc_traffic_shards_json.py $O/traffic_shards 3 > $O/pmc_traffic_shards_json.log 2>&1
cp profiles/r03_pmc_traffic.json profiles/r03_pmc_fetch_counter_collection.csv profiles/r03_pmc_traffic_shards.json $O/ 2>/dev/null
for n in 21 22 23; do CGPS_BENCH_FORCE_SHARDED=1 python bench.py --rows $((1<<n)) --steps 200 --warmup 20 --no-cpu-baseline > $O/shard_2p$n.json 2> $O/shard_2p$n.err; done
python bench.py --rows $((1<<24)) --steps 50 --warmup 5 --no-extras --no-cpu-baseline > $O/whole_2p24.json 2>/dev/null
CGPS_BENCH_REHEARSAL_GLOO=1 CGPS_BENCH_PREWARM_STEPS=50 python3 bench.py --gpus 2 --rows 4194304 --steps 20 --warmup 5 > $O/rehearsal_gloo_2ranks.json 2> $O/rehearsal.err; echo "rehearsal rc=$?"
bash tools/kstats.sh mahal_and_det --rows 4194304 --d 8 --dtype f32 --reps 40 > $O/c3_kstats.txt 2>&1
cat $O/pmc_traffic_json.log $O/pmc_traffic_shards_json.log
python - <<'PY'
import json
for f in ('bench_driver','bench_default','shard_2p21','shard_2p22','shard_2p23','whole_2p24','rehearsal_gloo_2ranks'):
    try:
        d=json.load(open('gpurun_out/r03p/%s.json'%f)); print(f, '%.2f us'%(d['ms_per_step']*1e3), 'frac %.4f' % d['roofline_frac_whole_op'], 'kernel %.1f' % d['roofline']['kernel_avg_us'], 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_stale'))
    except Exception as e: print(f, 'ERR', e)
PY
