#!/bin/bash
# Round-2 measurement set on the GPU box (writes under gpurun_out/r02p_*): headline kernel table of the
# driver's command, PMC traffic of the headline and of solve with 1 and 8 right-hand sides, config 3,
# the per-GPU part of the 8-GPU run (one 2^21-row shard through the sharded code path).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02p
# the two bench lines first (they also bring the GPU out of its idle power state for what follows)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02p/bench_driver.json 2> gpurun_out/r02p/bench_driver.err
python bench.py > gpurun_out/r02p/bench_default.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o p --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/r02p/headline_prof.log 2>&1; cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r02p/kernel_stats_headline.csv )
bash tools/pmc_traffic.sh gpurun_out/r02p/traffic_headline
for m in 1 8; do bash tools/pmc_traffic_op.sh gpurun_out/r02p/traffic_solve_m$m solve --nrhs $m; done
python tools/pmc_traffic_json.py gpurun_out/r02p/traffic_headline 2 > gpurun_out/r02p/pmc_traffic_json.log 2>&1   # (writes profiles/r02_pmc_traffic.json on the box: copied back below)
cp profiles/r02_pmc_traffic.json profiles/r02_pmc_fetch_counter_collection.csv gpurun_out/r02p/ 2>/dev/null
python tools/pmc_summary.py gpurun_out/r02p/traffic_headline gpurun_out/r02p/traffic_solve_m1 gpurun_out/r02p/traffic_solve_m8 > gpurun_out/r02p/pmc_summary.txt 2>&1
for m in 1 2 4 8; do python tools/prof_case.py --op solve --nrhs $m --reps 20; done > gpurun_out/r02p/solve_nrhs_times.txt 2>&1
bash tools/kstats.sh mahal_and_det --rows 4194304 --d 8 --dtype f32 --reps 40 > gpurun_out/r02p/c3_kstats.txt 2>&1
bash tools/kstats.sh inverse_blocks --rows 4194304 --d 8 --dtype f32 > gpurun_out/r02p/c3_inverse_kstats.txt 2>&1
CGPS_BENCH_FORCE_SHARDED=1 python bench.py --rows $((1<<21)) --steps 200 --warmup 20 > gpurun_out/r02p/shard_2p21.json 2> gpurun_out/r02p/shard_2p21.err
python bench.py --rows $((1<<21)) --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/r02p/whole_2p21.json 2>/dev/null
python bench.py --rows $((1<<24)) --steps 50 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r02p/whole_2p24.json 2>/dev/null
cat gpurun_out/r02p/pmc_summary.txt gpurun_out/r02p/solve_nrhs_times.txt gpurun_out/r02p/c3_kstats.txt
python -c "
import json
for f in ('shard_2p21','whole_2p21','whole_2p24'):
    d=json.load(open('gpurun_out/r02p/%s.json'%f)); print(f, '%.1f us'%(d['ms_per_step']*1e3))"
