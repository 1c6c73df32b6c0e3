#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_long_chunks.py tests/test_two_rounds.py tests/test_hip_parity.py tests/test_sharded.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for n in $((1<<20)) $((1<<21)) $((1<<23)) $((1<<24)) 8400001; do python tools/time_mahal.py $n 2>/dev/null; done
python tools/time_mahal.py 1500001 2 2>/dev/null; python tools/time_mahal.py 5600003 5 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('headline', d['ms_per_step']*1e3)"
