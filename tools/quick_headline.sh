#!/bin/bash
# Fast loop for the headline op on the GPU box: d=4 parity subset, the driver-protocol bench line,
# a 200-step line and the per-kernel durations.  Output under gpurun_out/ with the given tag.
tag=${1:-q}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_hip_parity.py tests/test_sharded.py -m gpu -x -q \
  -k "golden or (ragged_sizes and 4) or full_size or shard or stale" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/${tag}_tests.log
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null > gpurun_out/${tag}_bench20_$i.json
python -c "import json;d=json.load(open('gpurun_out/${tag}_bench20_$i.json'));print('20-step: %.2f us  frac %.4f  kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline_frac_whole_op'], d['roofline']['kernel_avg_us']))"
done
python bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null > gpurun_out/${tag}_bench200.json
python -c "import json;d=json.load(open('gpurun_out/${tag}_bench200.json'));print('200-step: %.2f us  frac %.4f  kernel %.1f us' % (d['ms_per_step']*1e3, d['roofline_frac_whole_op'], d['roofline']['kernel_avg_us']))"
bash tools/kstats.sh mahal_and_det
cp gpurun_out/kstats_mahal_and_det.csv gpurun_out/${tag}_kstats.csv
