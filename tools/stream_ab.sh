#!/bin/bash
# A/B of the persistent form (cgps_tile_stream.h) against rounds of chunk_reduce_kernel (CGPS_NO_STREAM=1):
# mahal_and_det at 2^20+4096 .. 2^24 rows, and the per-rank part of the 8-GPU run (one 2^21-row shard).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in $((1<<20)) $(((1<<20)+4096)) $((3<<19)) $((1<<21)) $((1<<22)) $((1<<23)) $((1<<24)); do
  a=$(python tools/prof_case.py --op mahal_and_det --rows $n --reps 100 | grep -o "[0-9.]* us per call")
  b=$(CGPS_NO_STREAM=1 python tools/prof_case.py --op mahal_and_det --rows $n --reps 100 | grep -o "[0-9.]* us per call")
  echo "rows $n: persistent $a | rounds $b"
done
for e in "" "CGPS_NO_STREAM=1"; do
  env $e CGPS_BENCH_FORCE_SHARDED=1 python bench.py --rows $((1<<21)) --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('shard 2^21 [$e]: %.1f us per step, kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_avg_us']))"
done
