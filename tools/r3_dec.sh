#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_autograd.py -m gpu -x -q -k "golden or ragged or full_size or inverse or autograd" > gpurun_out/r03c_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03c_tests.log
for d in 6 7 8; do
  python tools/prof_case.py --op decompose --d $d --reps 10 2>/dev/null
done
python tools/prof_case.py --op decompose --d 8 --rows 262144 --reps 10 2>/dev/null
python tools/prof_case.py --op decompose --d 7 --rows 300000 --reps 10 2>/dev/null
